"""hash_join_codes_knl_amd — MI355X (gfx950) hash-join hot path behind a C-ABI.

The product is libhjgpu.so (hand-written HIP kernels, include/hjgpu.h).  This
package holds its sources (csrc/), the C++ hosts that keep the reference's
./npj ./phj ./cpra ./write command lines (host/), the build driver and a
ctypes binding used by tests and bench.py.  No CPU fallback exists anywhere.
"""
from .api import (HjGpu, HjGpuError, DeviceColumn, NpjParams, PhjParams, Output, Result, Stats,
                  load_library, kernel_hash, library_hash, EXPORTS, FLAG_UNIQUE, HjComm, Shard, MultiStats, TRANSPORT_RCCL, TRANSPORT_LOOPBACK)
from . import build

__all__ = ["HjGpu", "HjGpuError", "DeviceColumn", "NpjParams", "PhjParams", "Output", "Result",
           "Stats", "load_library", "EXPORTS", "FLAG_UNIQUE", "HjComm", "Shard", "MultiStats",
           "TRANSPORT_RCCL", "TRANSPORT_LOOPBACK", "build"]

"""ctypes binding of the C-ABI in include/hjgpu.h — names, argument meaning and
error behaviour mirror the header one to one (which in turn cites the reference
operators each entry point replaces).

There is NO CPU fallback: if libhjgpu.so is missing or no GPU is visible the
binding raises.  Device memory is owned by the library (hjgpu_malloc) or by the
caller (any device pointer, e.g. torch.Tensor.data_ptr()).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OK, EINVAL, EALIGN, ENOMEM, EHIP, EZEROKEY, EOVERFLOW, ENODEVICE, ERCCL = range(9)
TRANSPORT_RCCL, TRANSPORT_LOOPBACK = 0, 1
MAX_FANOUT = 1024
MAX_PARTS = 32768
FLAG_UNIQUE = 1

EXPORTS = [
    "hjgpu_kernel_hash", "hjgpu_library_hash", "hjgpu_device_count", "hjgpu_create", "hjgpu_destroy", "hjgpu_last_error", "hjgpu_status_string",
    "hjgpu_get_device_info", "hjgpu_set_option", "hjgpu_reserve", "hjgpu_get_stats",
    "hjgpu_get_async_status", "hjgpu_accumulate_async_status", "hjgpu_set_async_output", "hjgpu_output_capacity",
    "hjgpu_malloc", "hjgpu_malloc_placed", "hjgpu_free", "hjgpu_memcpy_h2d", "hjgpu_memcpy_d2h", "hjgpu_synchronize", "hjgpu_audit_read", "hjgpu_audit_recheck",
    "hjgpu_host_alloc", "hjgpu_host_free",
    "hjgpu_histogram", "hjgpu_partition", "hjgpu_partition_async", "hjgpu_join_partitions",
    "hjgpu_npj_build", "hjgpu_npj_probe",
    "hjgpu_npj", "hjgpu_phj", "hjgpu_cpra",
    "hjgpu_npj_async", "hjgpu_phj_async", "hjgpu_cpra_async", "hjgpu_phj_overlapped_async",
    "hjgpu_phj_build", "hjgpu_phj_probe", "hjgpu_phj_probe_async",
    "hjgpu_partition_packed_async", "hjgpu_partition_packed_own_last_async", "hjgpu_phj_build_prepartitioned", "hjgpu_phj_probe_prepartitioned_async",
    "hjgpu_partition_packed_counted_async", "hjgpu_phj_probe_prepartitioned_counted_async", "hjgpu_prepartitioned_plan", "hjgpu_grouped_plan",
    "hjgpu_comm_create_local", "hjgpu_comm_get_id", "hjgpu_comm_create_rank", "hjgpu_comm_destroy",
    "hjgpu_comm_last_error", "hjgpu_comm_size", "hjgpu_comm_ctx", "hjgpu_comm_set_option", "hjgpu_comm_barrier",
    "hjgpu_comm_get_info", "hjgpu_comm_preflight", "hjgpu_comm_get_forensics", "hjgpu_comm_recheck", "hjgpu_comm_get_frozen",
    "hjgpu_phj_multi", "hjgpu_npj_multi", "hjgpu_cpra_multi", "hjgpu_join_host_multi",
    "hjgpu_phj_multi_rows", "hjgpu_npj_multi_rows", "hjgpu_cpra_multi_rows", "hjgpu_join_host_rows_multi",
    "hjgpu_join_host", "hjgpu_join_host_rows", "hjgpu_join_host_rows_shared", "hjgpu_generate", "hjgpu_generate_range", "hjgpu_generate_zipf", "hjgpu_generate_select", "hjgpu_column_sums", "hjgpu_stream_read_ms", "hjgpu_random_line_read_ms", "hjgpu_random_cas_ms",
]


class Result(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum_keys", C.c_uint64),
                ("sum_outer_vals", C.c_uint64), ("sum_inner_vals", C.c_uint64)]

    def as_tuple(self):
        return (self.count, self.sum_keys, self.sum_outer_vals, self.sum_inner_vals)


class Output(C.Structure):
    _fields_ = [("d_keys", C.c_void_p), ("d_outer_vals", C.c_void_p), ("d_inner_vals", C.c_void_p),
                ("capacity", C.c_size_t), ("block_size", C.c_size_t)]


class HostRows(C.Structure):
    _fields_ = [("keys", C.c_void_p), ("outer_vals", C.c_void_p), ("inner_vals", C.c_void_p),
                ("capacity", C.c_size_t)]


class PhjParams(C.Structure):
    _fields_ = [("fanout1", C.c_uint32), ("fanout2", C.c_uint32),
                ("factor1", C.c_uint32), ("factor2", C.c_uint32),
                ("table_factor", C.c_uint32 * 2),
                ("chunks", C.c_uint32), ("flags", C.c_uint32)]


class NpjParams(C.Structure):
    _fields_ = [("load", C.c_double), ("factor", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_histogram", C.c_float), ("ms_plan", C.c_float),
                ("ms_scatter1", C.c_float), ("ms_scatter2", C.c_float), ("ms_join", C.c_float),
                ("ms_build", C.c_float), ("ms_close_gaps", C.c_float),
                ("ms_inner_wait", C.c_float), ("ms_upload", C.c_float), ("ms_download", C.c_float),
                ("ms_reserve", C.c_float), ("fanout1", C.c_uint32), ("fanout2", C.c_uint32), ("batches", C.c_uint32), ("buckets", C.c_uint64),
                ("ms_scatter0", C.c_float), ("groups", C.c_uint32),
                ("placement_tried", C.c_uint32), ("placement_timeboxed", C.c_uint32), ("placement_fill_ms", C.c_float), ("placement_search_ms", C.c_float),
                ("placement_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Shard(C.Structure):
    """hjgpu_shard: one rank's share of the relations (device pointers on that rank's device)."""
    _fields_ = [("d_inner_keys", C.c_void_p), ("d_inner_vals", C.c_void_p), ("inner", C.c_size_t),
                ("d_outer_keys", C.c_void_p), ("d_outer_vals", C.c_void_p), ("outer", C.c_size_t)]


class MultiStats(C.Structure):
    _fields_ = [("ms_wall", C.c_float), ("ms_exchange", C.c_float), ("ms_partition", C.c_float),
                ("ms_exchange_wait", C.c_float), ("joins", C.c_uint32), ("self_copies", C.c_uint32),
                ("tuples_joined", C.c_uint64), ("bytes_sent", C.c_uint64), ("ms_upload", C.c_float),
                ("ms_overlap", C.c_float), ("join", Stats)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("join", "reserved")}
        d["join"] = self.join.as_dict()
        return d


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class CommInfo(C.Structure):
    """hjgpu_comm_info: the transport's own view of the world (ncclCommCount / ncclCommUserRank / ncclGetVersion)."""
    _fields_ = [("nranks", C.c_int), ("nlocal", C.c_int), ("first_rank", C.c_int), ("transport", C.c_char * 16),
                ("rccl_version", C.c_int), ("rccl_nranks", C.c_int), ("rccl_rank", C.c_int), ("rccl_device", C.c_int),
                ("timeout_ms", C.c_int), ("aborted", C.c_int)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["transport"] = d["transport"].decode()
        return d


class Preflight(C.Structure):
    _fields_ = [("nranks", C.c_uint32), ("rank", C.c_uint32), ("ok_all_gather", C.c_uint32), ("ok_all_to_all", C.c_uint32),
                ("ok_all_reduce", C.c_uint32), ("ms_all_gather", C.c_float), ("ms_all_to_all", C.c_float),
                ("ms_all_reduce", C.c_float), ("link_bytes", C.c_uint64), ("link_GBs", C.c_float * 64),
                ("all_to_all_GBs", C.c_float)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "link_GBs"}
        d["link_GBs"] = [round(float(x), 2) for x in self.link_GBs[:self.nranks]]
        return d


class PrePartitioned(C.Structure):
    """hjgpu_prepartitioned: a relation that arrives pass-1-partitioned in `chunks` pieces (multi-GPU CPRA receiver)."""
    _fields_ = [("factor1", C.c_uint32), ("fanout1_total", C.c_uint32), ("first_partition", C.c_uint32),
                ("fanout1", C.c_uint32), ("chunks", C.c_uint32), ("reserved", C.c_uint32), ("chunk_offsets", C.c_uint64 * 9)]


class ShardRows(C.Structure):
    """hjgpu_shard_rows: one rank's result columns and, after the call, its number of dense rows."""
    _fields_ = [("out", Output), ("rows", C.c_uint64)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int),
                ("lds_bytes_per_block", C.c_int), ("hbm_bytes", C.c_uint64)]


class PinnedColumn:
    """numpy view of page-locked host memory owned by the library (hjgpu_host_alloc)."""

    def __init__(self, hj, ptr, array):
        self.hj, self.ptr, self.array = hj, ptr, array

    def free(self):
        if self.ptr:
            self.array = None
            self.hj._check(self.hj.lib.hjgpu_host_free(self.hj.handle, self.ptr))
            self.ptr = None


class HjGpuError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("hjgpu status %d (%s)" % (status, text))
        self.status = status


_lib = None


def load_library(build_if_missing=True):
    """Loads hash_join_codes_knl_amd/lib/libhjgpu.so; raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    so = _build.lib_path()
    if not os.path.exists(so):
        if not build_if_missing:
            raise FileNotFoundError(so + " is missing: run `python -m hash_join_codes_knl_amd.build`")
        _build.build_library(verbose=False)
    L = C.CDLL(so)
    vp, sz, u32, u64p = C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64)
    L.hjgpu_kernel_hash.restype = C.c_char_p
    L.hjgpu_kernel_hash.argtypes = []
    L.hjgpu_library_hash.restype = C.c_char_p
    L.hjgpu_library_hash.argtypes = []
    L.hjgpu_device_count.argtypes = [C.POINTER(C.c_int)]
    L.hjgpu_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.hjgpu_destroy.argtypes = [vp]
    L.hjgpu_last_error.restype = C.c_char_p
    L.hjgpu_last_error.argtypes = [vp]
    L.hjgpu_status_string.restype = C.c_char_p
    L.hjgpu_status_string.argtypes = [C.c_int]
    L.hjgpu_get_device_info.argtypes = [vp, C.POINTER(DeviceInfo)]
    L.hjgpu_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.hjgpu_reserve.argtypes = [vp, sz, sz]
    L.hjgpu_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.hjgpu_get_async_status.argtypes = [vp, vp]
    L.hjgpu_accumulate_async_status.argtypes = [vp, vp, vp]
    L.hjgpu_set_async_output.argtypes = [vp, C.POINTER(Output)]
    L.hjgpu_output_capacity.argtypes = [vp, C.c_int, sz, sz, sz, C.POINTER(sz)]
    L.hjgpu_malloc.argtypes = [vp, C.POINTER(vp), sz]
    L.hjgpu_malloc_placed.argtypes = [vp, C.POINTER(vp), sz]
    L.hjgpu_free.argtypes = [vp, vp]
    L.hjgpu_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.hjgpu_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.hjgpu_synchronize.argtypes = [vp, vp]
    L.hjgpu_stream_read_ms.argtypes = [vp, vp, sz, C.POINTER(C.c_float), vp]
    L.hjgpu_random_line_read_ms.argtypes = [vp, vp, sz, sz, C.POINTER(C.c_float), vp]
    L.hjgpu_random_cas_ms.argtypes = [vp, vp, sz, sz, C.c_int, C.c_int, C.POINTER(C.c_float), vp]
    L.hjgpu_host_alloc.argtypes = [vp, C.POINTER(vp), sz]
    L.hjgpu_host_free.argtypes = [vp, vp]
    L.hjgpu_histogram.argtypes = [vp, vp, sz, u32, u32, vp, vp]
    L.hjgpu_partition.argtypes = [vp, vp, vp, sz, u32, u32, vp, vp, vp, vp]
    L.hjgpu_partition_async.argtypes = [vp, vp, vp, sz, u32, u32, vp, vp, vp, vp]
    L.hjgpu_comm_create_local.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.hjgpu_comm_get_id.argtypes = [C.POINTER(CommId)]
    L.hjgpu_comm_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(CommId), C.POINTER(vp)]
    L.hjgpu_comm_destroy.argtypes = [vp]
    L.hjgpu_comm_last_error.restype = C.c_char_p
    L.hjgpu_comm_last_error.argtypes = [vp]
    L.hjgpu_comm_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hjgpu_comm_ctx.restype = vp
    L.hjgpu_comm_ctx.argtypes = [vp, C.c_int]
    L.hjgpu_comm_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.hjgpu_comm_barrier.argtypes = [vp]
    L.hjgpu_comm_get_info.argtypes = [vp, C.POINTER(CommInfo)]
    L.hjgpu_comm_preflight.argtypes = [vp, sz, C.POINTER(Preflight)]
    L.hjgpu_comm_get_forensics.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(sz)]
    L.hjgpu_audit_read.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint64, u32, C.POINTER(C.c_uint64), vp]
    L.hjgpu_audit_recheck.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(sz)]
    L.hjgpu_comm_recheck.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(sz)]
    L.hjgpu_comm_get_frozen.argtypes = [vp, C.POINTER(C.c_uint64), sz, C.POINTER(sz)]
    L.hjgpu_phj_multi_rows.argtypes = [vp, C.POINTER(Shard), C.POINTER(ShardRows), C.c_int, C.POINTER(PhjParams), C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_npj_multi_rows.argtypes = [vp, C.POINTER(Shard), C.POINTER(ShardRows), C.c_int, C.POINTER(NpjParams), C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_cpra_multi_rows.argtypes = [vp, C.POINTER(Shard), C.POINTER(ShardRows), C.POINTER(PhjParams), C.c_int, C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_join_host_rows_multi.argtypes = [vp, C.c_int, vp, vp, sz, vp, vp, sz, C.POINTER(PhjParams), C.POINTER(NpjParams),
                                             C.POINTER(HostRows), C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_phj_multi.argtypes = [vp, C.POINTER(Shard), C.c_int, C.POINTER(PhjParams), C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_npj_multi.argtypes = [vp, C.POINTER(Shard), C.c_int, C.POINTER(NpjParams), C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_cpra_multi.argtypes = [vp, C.POINTER(Shard), C.POINTER(PhjParams), C.c_int, C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_join_host_multi.argtypes = [vp, C.c_int, vp, vp, sz, vp, vp, sz, C.POINTER(PhjParams), C.POINTER(NpjParams),
                                        C.POINTER(Result), C.POINTER(MultiStats)]
    L.hjgpu_join_partitions.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(PhjParams),
                                        C.POINTER(Result), C.POINTER(Output), vp]
    L.hjgpu_npj_build.argtypes = [vp, vp, vp, sz, vp, sz, u32, vp]
    L.hjgpu_npj_probe.argtypes = [vp, vp, vp, sz, vp, sz, u32, C.POINTER(Result),
                                  C.POINTER(Output), vp]
    join = [vp, vp, vp, sz, vp, vp, sz]
    L.hjgpu_npj.argtypes = join + [C.POINTER(NpjParams), C.POINTER(Result), C.POINTER(Output), vp]
    L.hjgpu_phj.argtypes = join + [C.POINTER(PhjParams), C.POINTER(Result), C.POINTER(Output), vp]
    L.hjgpu_cpra.argtypes = join + [C.POINTER(PhjParams), C.POINTER(Result), C.POINTER(Output), vp]
    L.hjgpu_npj_async.argtypes = join + [C.POINTER(NpjParams), vp, vp]
    L.hjgpu_phj_async.argtypes = join + [C.POINTER(PhjParams), vp, vp]
    L.hjgpu_cpra_async.argtypes = join + [C.POINTER(PhjParams), vp, vp]
    L.hjgpu_phj_overlapped_async.argtypes = join + [C.POINTER(PhjParams), vp, vp, vp]
    L.hjgpu_phj_build.argtypes = [vp, vp, vp, sz, sz, C.POINTER(PhjParams), vp]
    L.hjgpu_phj_probe.argtypes = [vp, vp, vp, sz, C.POINTER(Result), C.POINTER(Output), vp]
    L.hjgpu_phj_probe_async.argtypes = [vp, vp, vp, sz, vp, vp]
    L.hjgpu_partition_packed_async.argtypes = [vp, vp, vp, sz, u32, u32, vp, vp, vp]
    L.hjgpu_partition_packed_own_last_async.argtypes = [vp, vp, vp, sz, u32, u32, u32, u32, vp, vp, vp]
    L.hjgpu_partition_packed_counted_async.argtypes = [vp, vp, vp, sz, u32, u32, u32, u32, u32, u32, vp, vp, vp, vp]
    L.hjgpu_phj_probe_prepartitioned_counted_async.argtypes = [vp, vp, C.POINTER(PrePartitioned), vp, vp, vp]
    L.hjgpu_prepartitioned_plan.argtypes = [vp, sz, u32, C.POINTER(PhjParams), C.POINTER(u32), C.POINTER(u32)]
    L.hjgpu_grouped_plan.argtypes = [vp, sz, sz, C.POINTER(PhjParams), C.POINTER(u32)]
    L.hjgpu_phj_build_prepartitioned.argtypes = [vp, vp, C.POINTER(PrePartitioned), sz, C.POINTER(PhjParams), vp]
    L.hjgpu_phj_probe_prepartitioned_async.argtypes = [vp, vp, C.POINTER(PrePartitioned), vp, vp]
    L.hjgpu_join_host.argtypes = [vp, C.c_int, vp, vp, sz, vp, vp, sz, C.POINTER(PhjParams),
                                  C.POINTER(NpjParams), C.POINTER(Result), C.POINTER(Stats)]
    L.hjgpu_join_host_rows.argtypes = [vp, C.c_int, vp, vp, sz, vp, vp, sz, C.POINTER(PhjParams),
                                       C.POINTER(NpjParams), C.POINTER(HostRows), C.POINTER(Result), C.POINTER(Stats)]
    L.hjgpu_join_host_rows_shared.argtypes = [vp, C.c_int, vp, vp, sz, vp, vp, sz, C.POINTER(PhjParams),
                                              C.POINTER(NpjParams), C.POINTER(HostRows), u64p, C.POINTER(Result), C.POINTER(Stats)]
    L.hjgpu_generate.argtypes = [vp, C.c_uint64, sz, sz, sz, sz, u32, u32, vp, vp, vp, vp, vp]
    L.hjgpu_generate_range.argtypes = [vp, C.c_uint64, sz, sz, sz, sz, sz, sz, u32, u32, vp, vp, vp, vp, vp]
    L.hjgpu_generate_zipf.argtypes = [vp, C.c_uint64, sz, sz, sz, sz, sz, sz, u32, u32, C.c_double, vp, vp, vp, vp, vp]
    L.hjgpu_generate_select.argtypes = [vp, C.c_uint64, sz, sz, sz, sz, sz, sz, u32, u32, C.c_double, C.c_double,
                                        vp, vp, vp, vp, C.POINTER(Result), vp]
    L.hjgpu_column_sums.argtypes = [vp, vp, sz, u32, u32, u64p, vp]
    for name in EXPORTS:
        if name not in ("hjgpu_last_error", "hjgpu_status_string", "hjgpu_comm_last_error", "hjgpu_comm_ctx", "hjgpu_kernel_hash", "hjgpu_library_hash"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def library_hash():
    """hjgpu_library_hash(): every source the loaded library was built from (evidence headers)."""
    return load_library().hjgpu_library_hash().decode()


def kernel_hash():
    """hjgpu_kernel_hash(): which kernel sources the loaded library was built from."""
    return load_library().hjgpu_kernel_hash().decode()


class DeviceColumn:
    """A uint32/uint64 column in HBM owned through hjgpu_malloc."""

    def __init__(self, ctx, n, dtype=np.uint32, placed=False):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        p = C.c_void_p()
        alloc = ctx.lib.hjgpu_malloc_placed if placed else ctx.lib.hjgpu_malloc
        ctx._check(alloc(ctx.handle, C.byref(p), self.n * self.dtype.itemsize))
        self.ptr = p.value

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.n
        self.ctx._check(self.ctx.lib.hjgpu_memcpy_h2d(self.ctx.handle, self.ptr,
                                                      host.ctypes.data, self.nbytes))
        return self

    def download(self, n=None):
        n = self.n if n is None else int(n)
        host = np.empty(n, self.dtype)
        self.ctx._check(self.ctx.lib.hjgpu_memcpy_d2h(self.ctx.handle, host.ctypes.data,
                                                      self.ptr, n * self.dtype.itemsize))
        return host

    def free(self):
        if self.ptr:
            self.ctx.lib.hjgpu_free(self.ctx.handle, self.ptr)
            self.ptr = None


class HjGpu:
    """One hjgpu_ctx.  Methods are named after the C entry points (hjgpu_ prefix dropped)."""

    def __init__(self, device=-1, _borrowed=None):
        self.lib = load_library()
        self._owned = _borrowed is None
        if _borrowed is not None:          # a communicator's context (hjgpu_comm_ctx): destroyed with the communicator
            self.handle = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        st = self.lib.hjgpu_create(device, C.byref(h))
        if st != OK:
            raise HjGpuError(st, self.lib.hjgpu_status_string(st).decode())
        self.handle = h

    def close(self):
        if self.handle and self._owned:
            self.lib.hjgpu_destroy(self.handle)
        self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != OK:
            raise HjGpuError(st, "%s: %s" % (self.lib.hjgpu_status_string(st).decode(),
                                             self.lib.hjgpu_last_error(self.handle).decode()))

    # ---- helpers -----------------------------------------------------------------
    def column(self, host_or_n, dtype=np.uint32, placed=False):
        """placed: hjgpu_malloc_placed (result columns of a gigabyte and more: the placement search of the workspace)"""
        if isinstance(host_or_n, (int, np.integer)):
            return DeviceColumn(self, host_or_n, dtype, placed)
        host = np.ascontiguousarray(host_or_n, dtype=dtype)
        return DeviceColumn(self, host.size, dtype).upload(host)

    def device_info(self):
        info = DeviceInfo()
        self._check(self.lib.hjgpu_get_device_info(self.handle, C.byref(info)))
        return {"name": info.name.decode(), "arch": info.arch.decode(),
                "compute_units": info.compute_units,
                "lds_bytes_per_block": info.lds_bytes_per_block, "hbm_bytes": info.hbm_bytes}

    def set_option(self, name, value):
        """hjgpu_set_option: tuning / test switch of this context ("unique", "force_chained", "dense2", ...)."""
        if isinstance(value, bool):
            value = int(value)
        self._check(self.lib.hjgpu_set_option(self.handle, name.encode(), str(value).encode()))

    def reserve(self, inner, outer):
        self._check(self.lib.hjgpu_reserve(self.handle, inner, outer))

    def stats(self):
        s = Stats()
        self._check(self.lib.hjgpu_get_stats(self.handle, C.byref(s)))
        return s.as_dict()

    def synchronize(self, stream=None):
        self._check(self.lib.hjgpu_synchronize(self.handle, stream))

    def get_async_status(self, stream=None):
        """hjgpu_get_async_status: raises what the blocking form of the last *_async join would have raised
        (HJGPU_EZEROKEY, HJGPU_EOVERFLOW) after waiting for `stream`."""
        self._check(self.lib.hjgpu_get_async_status(self.handle, stream))

    def accumulate_async_status(self, d_flags, stream=None):
        self._check(self.lib.hjgpu_accumulate_async_status(self.handle, self._ptr(d_flags), stream))

    def set_async_output(self, out):
        """hjgpu_set_async_output: the next *_async join of this context materialises into out = (keys, outer_vals,
        inner_vals, capacity, block_size); None withdraws it."""
        self._check(self.lib.hjgpu_set_async_output(self.handle, self._out(out)))

    def output_capacity(self, algorithm, outer, rows, block_size=0):
        cap = C.c_size_t()
        self._check(self.lib.hjgpu_output_capacity(self.handle, algorithm, outer, rows, block_size, C.byref(cap)))
        return cap.value

    @staticmethod
    def _ptr(x):
        return x.ptr if isinstance(x, DeviceColumn) else x

    def _out(self, out):
        if out is None:
            return None
        keys, ov, iv, capacity, block_size = out
        return C.byref(Output(self._ptr(keys), self._ptr(ov), self._ptr(iv), capacity, block_size))

    # ---- operators ------------------------------------------------------------------
    def histogram(self, d_keys, n, factor, fanout, d_counts, stream=None):
        self._check(self.lib.hjgpu_histogram(self.handle, self._ptr(d_keys), n, factor, fanout,
                                             self._ptr(d_counts), stream))

    def partition_async(self, d_keys, d_vals, n, factor, fanout, d_keys_out, d_vals_out, d_offsets, stream=None):
        self._check(self.lib.hjgpu_partition_async(self.handle, self._ptr(d_keys), self._ptr(d_vals), n,
                                                   factor, fanout, self._ptr(d_keys_out),
                                                   self._ptr(d_vals_out), self._ptr(d_offsets), stream))

    def partition(self, d_keys, d_vals, n, factor, fanout, d_keys_out, d_vals_out, d_offsets,
                  stream=None):
        self._check(self.lib.hjgpu_partition(self.handle, self._ptr(d_keys), self._ptr(d_vals), n,
                                             factor, fanout, self._ptr(d_keys_out),
                                             self._ptr(d_vals_out), self._ptr(d_offsets), stream))

    def join_partitions(self, rk, rv, roff, sk, sv, soff, params, out=None, stream=None):
        r = Result()
        self._check(self.lib.hjgpu_join_partitions(self.handle, self._ptr(rk), self._ptr(rv),
                                                   self._ptr(roff), self._ptr(sk), self._ptr(sv),
                                                   self._ptr(soff), C.byref(params), C.byref(r),
                                                   self._out(out), stream))
        return r.as_tuple()

    def npj_build(self, d_keys, d_vals, n, d_table, buckets, factor, stream=None):
        self._check(self.lib.hjgpu_npj_build(self.handle, self._ptr(d_keys), self._ptr(d_vals), n,
                                             self._ptr(d_table), buckets, factor, stream))

    def npj_probe(self, d_keys, d_vals, n, d_table, buckets, factor, out=None, stream=None):
        r = Result()
        self._check(self.lib.hjgpu_npj_probe(self.handle, self._ptr(d_keys), self._ptr(d_vals), n,
                                             self._ptr(d_table), buckets, factor, C.byref(r),
                                             self._out(out), stream))
        return r.as_tuple()

    # ---- whole joins ------------------------------------------------------------------
    def _join(self, fn, params, rk, rv, inner, sk, sv, outer, out, stream):
        r = Result()
        self._check(fn(self.handle, self._ptr(rk), self._ptr(rv), inner, self._ptr(sk),
                       self._ptr(sv), outer, C.byref(params) if params is not None else None,
                       C.byref(r), self._out(out), stream))
        return r.as_tuple()

    def npj(self, rk, rv, inner, sk, sv, outer, params=None, out=None, stream=None):
        return self._join(self.lib.hjgpu_npj, params, rk, rv, inner, sk, sv, outer, out, stream)

    def phj(self, rk, rv, inner, sk, sv, outer, params=None, out=None, stream=None):
        return self._join(self.lib.hjgpu_phj, params, rk, rv, inner, sk, sv, outer, out, stream)

    def cpra(self, rk, rv, inner, sk, sv, outer, params=None, out=None, stream=None):
        return self._join(self.lib.hjgpu_cpra, params, rk, rv, inner, sk, sv, outer, out, stream)

    # build side prepared once, probed by batches (results are per batch)
    def phj_build(self, rk, rv, inner, max_outer, params=None, stream=None):
        self._check(self.lib.hjgpu_phj_build(self.handle, self._ptr(rk), self._ptr(rv), inner, max_outer,
                                             C.byref(params) if params is not None else None, stream))

    def phj_probe(self, sk, sv, outer, out=None, stream=None):
        r = Result()
        self._check(self.lib.hjgpu_phj_probe(self.handle, self._ptr(sk), self._ptr(sv), outer, C.byref(r),
                                             self._out(out), stream))
        return r.as_tuple()

    def phj_probe_async(self, sk, sv, outer, d_result, stream=None):
        self._check(self.lib.hjgpu_phj_probe_async(self.handle, self._ptr(sk), self._ptr(sv), outer,
                                                   self._ptr(d_result), stream))

    # relations that arrive pass-1-partitioned (the receiving side of the multi-GPU CPRA)
    def partition_packed_async(self, d_keys, d_vals, n, factor, fanout, d_tuples_out, d_offsets, stream=None):
        self._check(self.lib.hjgpu_partition_packed_async(self.handle, self._ptr(d_keys), self._ptr(d_vals), n, factor, fanout,
                                                          self._ptr(d_tuples_out), self._ptr(d_offsets), stream))

    def partition_packed_own_last_async(self, d_keys, d_vals, n, factor, fanout, own_first, own_count, d_tuples_out, d_offsets, stream=None):
        self._check(self.lib.hjgpu_partition_packed_own_last_async(self.handle, self._ptr(d_keys), self._ptr(d_vals), n, factor, fanout,
                                                                   own_first, own_count, self._ptr(d_tuples_out), self._ptr(d_offsets), stream))

    def partition_packed_counted_async(self, d_keys, d_vals, n, factor, fanout, own_first, own_count, factor2, fanout2,
                                       d_tuples_out, d_offsets, d_counts2, stream=None):
        self._check(self.lib.hjgpu_partition_packed_counted_async(self.handle, self._ptr(d_keys), self._ptr(d_vals), n, factor, fanout,
                                                                  own_first, own_count, factor2, fanout2, self._ptr(d_tuples_out),
                                                                  self._ptr(d_offsets), self._ptr(d_counts2), stream))

    def prepartitioned_plan(self, inner, fanout1, params=None):
        """(fanout2, factor2) that hjgpu_phj_build_prepartitioned plans for `inner` build rows in `fanout1` pass-1 partitions"""
        f2, m2 = C.c_uint32(), C.c_uint32()
        self._check(self.lib.hjgpu_prepartitioned_plan(self.handle, inner, fanout1, C.byref(params) if params is not None else None,
                                                       C.byref(f2), C.byref(m2)))
        return f2.value, m2.value

    def phj_probe_prepartitioned_counted_async(self, d_tuples, layout, d_counts, d_result, stream=None):
        self._check(self.lib.hjgpu_phj_probe_prepartitioned_counted_async(self.handle, self._ptr(d_tuples), C.byref(layout),
                                                                          self._ptr(d_counts), self._ptr(d_result), stream))

    @staticmethod
    def prepartitioned(factor1, fanout1_total, first_partition, fanout1, chunk_offsets):
        lay = PrePartitioned(factor1, fanout1_total, first_partition, fanout1, len(chunk_offsets) - 1, 0)
        for i in range(9):
            lay.chunk_offsets[i] = int(chunk_offsets[min(i, len(chunk_offsets) - 1)])
        return lay

    def phj_build_prepartitioned(self, d_tuples, layout, max_outer, params=None, stream=None):
        self._check(self.lib.hjgpu_phj_build_prepartitioned(self.handle, self._ptr(d_tuples), C.byref(layout), max_outer,
                                                            C.byref(params) if params is not None else None, stream))

    def phj_probe_prepartitioned_async(self, d_tuples, layout, d_result, stream=None):
        self._check(self.lib.hjgpu_phj_probe_prepartitioned_async(self.handle, self._ptr(d_tuples), C.byref(layout),
                                                                  self._ptr(d_result), stream))

    def _join_async(self, fn, params, rk, rv, inner, sk, sv, outer, d_result, stream):
        self._check(fn(self.handle, self._ptr(rk), self._ptr(rv), inner, self._ptr(sk),
                       self._ptr(sv), outer, C.byref(params) if params is not None else None,
                       self._ptr(d_result), stream))

    def npj_async(self, rk, rv, inner, sk, sv, outer, params, d_result, stream=None):
        self._join_async(self.lib.hjgpu_npj_async, params, rk, rv, inner, sk, sv, outer, d_result, stream)

    def phj_async(self, rk, rv, inner, sk, sv, outer, params, d_result, stream=None):
        self._join_async(self.lib.hjgpu_phj_async, params, rk, rv, inner, sk, sv, outer, d_result, stream)

    def phj_overlapped_async(self, rk, rv, inner, sk, sv, outer, params, d_result, stream, inner_ready_event):
        self._check(self.lib.hjgpu_phj_overlapped_async(
            self.handle, self._ptr(rk), self._ptr(rv), inner, self._ptr(sk), self._ptr(sv), outer,
            C.byref(params) if params is not None else None, self._ptr(d_result), stream, inner_ready_event))

    def cpra_async(self, rk, rv, inner, sk, sv, outer, params, d_result, stream=None):
        self._join_async(self.lib.hjgpu_cpra_async, params, rk, rv, inner, sk, sv, outer, d_result, stream)

    def host_column(self, n, dtype=np.uint32):
        """Page-locked host column (hjgpu_host_alloc) as a numpy array; `arr.base_free()` releases it."""
        p = C.c_void_p()
        nbytes = int(n) * np.dtype(dtype).itemsize
        self._check(self.lib.hjgpu_host_alloc(self.handle, C.byref(p), nbytes))
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(n))
        return PinnedColumn(self, p.value, arr)

    def join_host(self, algorithm, ik, iv, ok, ov, phj_params=None, npj_params=None):
        """algorithm: 0 npj, 1 phj, 2 cpra; host numpy columns in, (result, stats) out."""
        ik, iv, ok, ov = (c.array if isinstance(c, PinnedColumn) else np.ascontiguousarray(c, np.uint32)
                          for c in (ik, iv, ok, ov))
        r, s = Result(), Stats()
        self._check(self.lib.hjgpu_join_host(
            self.handle, algorithm, ik.ctypes.data, iv.ctypes.data, ik.size,
            ok.ctypes.data, ov.ctypes.data, ok.size,
            C.byref(phj_params) if phj_params is not None else None,
            C.byref(npj_params) if npj_params is not None else None, C.byref(r), C.byref(s)))
        return r.as_tuple(), s.as_dict()

    def join_host_rows(self, algorithm, ik, iv, ok, ov, capacity, phj_params=None, npj_params=None, pinned=False):
        """As join_host, with the join materialised into three host columns of `capacity` rows
        (page-locked if `pinned`): returns (result, stats, (keys, outer_vals, inner_vals)) with the
        columns cut to result.count rows.  Raises HjGpuError(HJGPU_EOVERFLOW) when the join has more
        rows than `capacity`."""
        ik, iv, ok, ov = (c.array if isinstance(c, PinnedColumn) else np.ascontiguousarray(c, np.uint32)
                          for c in (ik, iv, ok, ov))
        cols = [self.host_column(max(capacity, 1)) if pinned else np.empty(max(capacity, 1), np.uint32) for _ in range(3)]
        arrs = [c.array if pinned else c for c in cols]
        rows = HostRows(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, capacity)
        r, s = Result(), Stats()
        try:
            self._check(self.lib.hjgpu_join_host_rows(
                self.handle, algorithm, ik.ctypes.data, iv.ctypes.data, ik.size,
                ok.ctypes.data, ov.ctypes.data, ok.size,
                C.byref(phj_params) if phj_params is not None else None,
                C.byref(npj_params) if npj_params is not None else None, C.byref(rows), C.byref(r), C.byref(s)))
            out = tuple(a[:r.count].copy() for a in arrs)
        finally:
            if pinned:
                for c in cols:
                    c.free()
        return r.as_tuple(), s.as_dict(), out

    # ---- generator --------------------------------------------------------------------
    def generate(self, seed, inner, outer_total, outer_begin, outer_count, inner_factor,
                 outer_factor, ik, iv, ok, ov, stream=None):
        self._check(self.lib.hjgpu_generate(self.handle, seed, inner, outer_total, outer_begin,
                                            outer_count, inner_factor, outer_factor,
                                            self._ptr(ik), self._ptr(iv), self._ptr(ok),
                                            self._ptr(ov), stream))

    def generate_range(self, seed, inner_total, outer_total, inner_begin, inner_count, outer_begin,
                       outer_count, inner_factor, outer_factor, ik, iv, ok, ov, stream=None):
        self._check(self.lib.hjgpu_generate_range(self.handle, seed, inner_total, outer_total, inner_begin,
                                                  inner_count, outer_begin, outer_count, inner_factor,
                                                  outer_factor, self._ptr(ik), self._ptr(iv),
                                                  self._ptr(ok), self._ptr(ov), stream))

    def generate_zipf(self, seed, inner_total, outer_total, inner_begin, inner_count, outer_begin,
                      outer_count, inner_factor, outer_factor, zipf, ik, iv, ok, ov, stream=None):
        self._check(self.lib.hjgpu_generate_zipf(self.handle, seed, inner_total, outer_total, inner_begin,
                                                 inner_count, outer_begin, outer_count, inner_factor,
                                                 outer_factor, float(zipf), self._ptr(ik), self._ptr(iv),
                                                 self._ptr(ok), self._ptr(ov), stream))

    def generate_select(self, seed, inner_total, outer_total, inner_begin, inner_count, outer_begin, outer_count,
                        inner_factor, outer_factor, zipf, selectivity, ik, iv, ok, ov, expected=True, stream=None):
        """hjgpu_generate_select: write.cpp's selectivity; returns the analytic aggregates of the generated probe
        range (None when expected is False)."""
        r = Result()
        self._check(self.lib.hjgpu_generate_select(self.handle, seed, inner_total, outer_total, inner_begin, inner_count,
                                                   outer_begin, outer_count, inner_factor, outer_factor, float(zipf),
                                                   float(selectivity), self._ptr(ik), self._ptr(iv), self._ptr(ok),
                                                   self._ptr(ov), C.byref(r) if expected else None, stream))
        return r.as_tuple() if expected else None

    def stream_read_ms(self, d_ptr, nbytes, stream=None):
        ms = C.c_float()
        self._check(self.lib.hjgpu_stream_read_ms(self.handle, self._ptr(d_ptr), nbytes, C.byref(ms), stream))
        return ms.value

    def random_cas_ms(self, d_ptr, nbytes, ops, in_flight=4, load_first=False, stream=None):
        """hjgpu_random_cas_ms: ms for `ops` independent random 8-byte CAS into the (zeroed) buffer: the NPJ build's ceiling"""
        ms = C.c_float()
        self._check(self.lib.hjgpu_random_cas_ms(self.handle, self._ptr(d_ptr), nbytes, ops, in_flight, int(bool(load_first)), C.byref(ms), stream))
        return ms.value

    def random_line_read_ms(self, d_ptr, nbytes, reads, stream=None):
        ms = C.c_float()
        self._check(self.lib.hjgpu_random_line_read_ms(self.handle, self._ptr(d_ptr), nbytes, reads, C.byref(ms), stream))
        return ms.value

    def column_sums(self, d_keys, n, fa, fb, stream=None):
        sums = (C.c_uint64 * 3)()
        self._check(self.lib.hjgpu_column_sums(self.handle, self._ptr(d_keys), n, fa, fb, sums, stream))
        return tuple(int(x) for x in sums)


class HjComm:
    """One hjgpu_comm: the ranks of a multi-GPU join that live in this process.

    HjComm.local(n, devices, transport): every rank in this process (RCCL: one rank per device; LOOPBACK: ranks may
    share a device, messages are device-to-device copies).  HjComm.rank(device, n, rank, id): one rank per process,
    `id` = HjComm.new_id() of rank 0, passed around by the launcher (bench.py: torch.distributed's gloo group)."""

    def __init__(self, handle):
        self.lib = load_library()
        self.handle = handle
        n, l, f = C.c_int(), C.c_int(), C.c_int()
        self.lib.hjgpu_comm_size(self.handle, C.byref(n), C.byref(l), C.byref(f))
        self.nranks, self.nlocal, self.first_rank = n.value, l.value, f.value
        self.ctx = [HjGpu(_borrowed=self.lib.hjgpu_comm_ctx(self.handle, i)) for i in range(self.nlocal)]

    @staticmethod
    def _create_error(lib, st):
        # the communicator that could not be made is gone: its text is kept per thread (hjgpu_comm_last_error(NULL))
        return HjGpuError(st, "%s: %s" % (lib.hjgpu_status_string(st).decode(), lib.hjgpu_comm_last_error(None).decode()))

    @classmethod
    def local(cls, nranks, devices=None, transport=TRANSPORT_RCCL):
        lib = load_library()
        h = C.c_void_p()
        devs = (C.c_int * max(nranks, 1))(*devices) if devices is not None else None
        st = lib.hjgpu_comm_create_local(nranks, devs, transport, C.byref(h))
        if st != OK:
            raise cls._create_error(lib, st)
        return cls(h)

    @staticmethod
    def new_id():
        lib = load_library()
        cid = CommId()
        st = lib.hjgpu_comm_get_id(C.byref(cid))
        if st != OK:
            raise HjComm._create_error(lib, st)
        return bytes(bytearray(C.string_at(C.addressof(cid), 128)))

    @classmethod
    def rank(cls, device, nranks, rank, comm_id):
        lib = load_library()
        h = C.c_void_p()
        cid = CommId()
        C.memmove(C.addressof(cid), comm_id, 128)
        st = lib.hjgpu_comm_create_rank(device, nranks, rank, C.byref(cid), C.byref(h))
        if st != OK:
            raise cls._create_error(lib, st)
        return cls(h)

    def close(self):
        if self.handle:
            self.lib.hjgpu_comm_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st):
        if st != OK:
            raise HjGpuError(st, "%s: %s" % (self.lib.hjgpu_status_string(st).decode(),
                                             self.lib.hjgpu_comm_last_error(self.handle).decode()))

    def set_option(self, name, value):
        self._check(self.lib.hjgpu_comm_set_option(self.handle, name.encode(), str(int(value)).encode()))

    def barrier(self):
        self._check(self.lib.hjgpu_comm_barrier(self.handle))

    def info(self):
        """hjgpu_comm_get_info: transport, RCCL version, ncclCommCount / ncclCommUserRank of local rank 0, timeout, aborted."""
        i = CommInfo()
        self._check(self.lib.hjgpu_comm_get_info(self.handle, C.byref(i)))
        return i.as_dict()

    def forensics(self):
        """hjgpu_comm_get_forensics (option debug_forensics): per local rank (global rank, partitioning records, join records);
        a record = 8 stages x [misplaced, sum of keys, sum of payloads, tuples] (include/hjgpu.h: hjgpu_audit_read)."""
        n = C.c_size_t(0)
        self._check(self.lib.hjgpu_comm_get_forensics(self.handle, None, 0, C.byref(n)))
        buf = (C.c_uint64 * max(1, n.value))()
        self._check(self.lib.hjgpu_comm_get_forensics(self.handle, buf, n.value, C.byref(n)))
        out, at = [], 0
        while at < n.value:
            rank, npart, njoin = int(buf[at]), int(buf[at + 1]), int(buf[at + 2])
            at += 3
            recs = []
            for _ in range(npart + njoin):
                recs.append([[int(buf[at + 4 * s + w]) for w in range(4)] for s in range(8)])
                at += 32
            out.append((rank, recs[:npart], recs[npart:]))
        return out

    def recheck(self):
        """hjgpu_comm_recheck: the last audited calls' partition checks again with the device quiet; per local rank and context
        (global rank, context 0 partitioning / 1 join, [(stage, [4 words as a fresh kernel counts], [4 words as the host counts from a hipMemcpy])])."""
        n = C.c_size_t(0)
        self._check(self.lib.hjgpu_comm_recheck(self.handle, None, 0, C.byref(n)))
        buf = (C.c_uint64 * max(1, n.value))()
        self._check(self.lib.hjgpu_comm_recheck(self.handle, buf, n.value, C.byref(n)))
        out, at = [], 0
        while at < n.value:
            rank, which, k = int(buf[at]), int(buf[at + 1]), int(buf[at + 2])
            at += 3
            checks = []
            for _ in range(k):
                checks.append((int(buf[at]), [int(buf[at + 1 + w]) for w in range(4)], [int(buf[at + 5 + w]) for w in range(4)]))
                at += 9
            out.append((rank, which, checks))
        return out

    def frozen(self):
        """hjgpu_comm_get_frozen (option debug_forensics = 2): the stages of the last step found wrong while their buffers were intact:
        [(global rank, context, slice, record[8][4], [(stage, fresh kernel's 4 words, host copy's 4 words)])]"""
        n = C.c_size_t(0)
        self._check(self.lib.hjgpu_comm_get_frozen(self.handle, None, 0, C.byref(n)))
        buf = (C.c_uint64 * max(1, n.value))()
        self._check(self.lib.hjgpu_comm_get_frozen(self.handle, buf, n.value, C.byref(n)))
        out, at = [], 0
        while at < n.value:
            rank, which, sl, k = int(buf[at]), int(buf[at + 1]), int(C.c_int64(buf[at + 2]).value), int(buf[at + 3])
            rec = [[int(buf[at + 4 + 4 * s + w]) for w in range(4)] for s in range(8)]
            at += 36
            checks = []
            for _ in range(k):
                checks.append((int(buf[at]), [int(buf[at + 1 + w]) for w in range(4)], [int(buf[at + 5 + w]) for w in range(4)]))
                at += 9
            out.append((rank, which, sl, rec, checks))
        return out

    def preflight(self, link_bytes=256 << 20):
        """hjgpu_comm_preflight: checksum-verified collectives + point-to-point rates (GB/s per peer)."""
        p = Preflight()
        self._check(self.lib.hjgpu_comm_preflight(self.handle, link_bytes, C.byref(p)))
        return p.as_dict()

    def _shards(self, shards):
        """shards: one (rk, rv, inner, sk, sv, outer) per local rank; columns are DeviceColumns / pointers / None."""
        assert len(shards) == self.nlocal
        arr = (Shard * self.nlocal)()
        for i, (rk, rv, inner, sk, sv, outer) in enumerate(shards):
            p = HjGpu._ptr
            arr[i] = Shard(p(rk), p(rv), inner, p(sk), p(sv), outer)
        return arr

    def _run(self, fn, shards, *mid):
        r, s = Result(), MultiStats()
        self._check(fn(self.handle, self._shards(shards), *mid, C.byref(r), C.byref(s)))
        return r.as_tuple(), s.as_dict()

    def phj_multi(self, shards, root=0, params=None):
        return self._run(self.lib.hjgpu_phj_multi, shards, root, C.byref(params) if params is not None else None)

    def npj_multi(self, shards, root=0, params=None):
        return self._run(self.lib.hjgpu_npj_multi, shards, root, C.byref(params) if params is not None else None)

    def cpra_multi(self, shards, params=None, slices=0):
        return self._run(self.lib.hjgpu_cpra_multi, shards, C.byref(params) if params is not None else None, slices)

    # ---- materialised rows: outs = one (keys, outer_vals, inner_vals, capacity, block_size) per local rank --------
    def _rows(self, outs):
        assert len(outs) == self.nlocal
        arr = (ShardRows * self.nlocal)()
        for i, (k, o, v, cap, bs) in enumerate(outs):
            p = HjGpu._ptr
            arr[i].out = Output(p(k), p(o), p(v), cap, bs)
        return arr

    def _run_rows(self, fn, shards, outs, *mid):
        """returns (result, stats, rows per local rank); on HJGPU_EOVERFLOW the exception carries .result and .rows
        (what every rank needs)"""
        r, s = Result(), MultiStats()
        rows = self._rows(outs)
        st = fn(self.handle, self._shards(shards), rows, *mid, C.byref(r), C.byref(s))
        counts = [int(rows[i].rows) for i in range(self.nlocal)]
        if st != OK:
            e = HjGpuError(st, "%s: %s" % (self.lib.hjgpu_status_string(st).decode(),
                                           self.lib.hjgpu_comm_last_error(self.handle).decode()))
            e.result, e.rows = r.as_tuple(), counts
            raise e
        return r.as_tuple(), s.as_dict(), counts

    def phj_multi_rows(self, shards, outs, root=0, params=None):
        return self._run_rows(self.lib.hjgpu_phj_multi_rows, shards, outs, root, C.byref(params) if params is not None else None)

    def npj_multi_rows(self, shards, outs, root=0, params=None):
        return self._run_rows(self.lib.hjgpu_npj_multi_rows, shards, outs, root, C.byref(params) if params is not None else None)

    def cpra_multi_rows(self, shards, outs, params=None, slices=0):
        return self._run_rows(self.lib.hjgpu_cpra_multi_rows, shards, outs, C.byref(params) if params is not None else None, slices)

    def join_host_rows_multi(self, algorithm, ik, iv, ok, ov, capacity, phj_params=None, npj_params=None):
        """hjgpu_join_host_rows_multi: (result, stats, (keys, outer_vals, inner_vals)) with the columns cut to count rows"""
        ik, iv, ok, ov = (np.ascontiguousarray(c, np.uint32) for c in (ik, iv, ok, ov))
        cols = [np.empty(max(capacity, 1), np.uint32) for _ in range(3)]
        rows = HostRows(cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, capacity)
        r, s = Result(), MultiStats()
        st = self.lib.hjgpu_join_host_rows_multi(
            self.handle, algorithm, ik.ctypes.data, iv.ctypes.data, ik.size, ok.ctypes.data, ov.ctypes.data, ok.size,
            C.byref(phj_params) if phj_params is not None else None,
            C.byref(npj_params) if npj_params is not None else None, C.byref(rows), C.byref(r), C.byref(s))
        if st != OK:
            e = HjGpuError(st, "%s: %s" % (self.lib.hjgpu_status_string(st).decode(),
                                           self.lib.hjgpu_comm_last_error(self.handle).decode()))
            e.result = r.as_tuple()
            raise e
        return r.as_tuple(), s.as_dict(), tuple(c[:r.count].copy() for c in cols)

    def join_host_multi(self, algorithm, ik, iv, ok, ov, phj_params=None, npj_params=None):
        ik, iv, ok, ov = (np.ascontiguousarray(c, np.uint32) for c in (ik, iv, ok, ov))
        r, s = Result(), MultiStats()
        self._check(self.lib.hjgpu_join_host_multi(
            self.handle, algorithm, ik.ctypes.data, iv.ctypes.data, ik.size, ok.ctypes.data, ov.ctypes.data, ok.size,
            C.byref(phj_params) if phj_params is not None else None,
            C.byref(npj_params) if npj_params is not None else None, C.byref(r), C.byref(s)))
        return r.as_tuple(), s.as_dict()

#!/usr/bin/env python3
"""Repeats hjgpu_cpra_multi on one communicator and reports every step whose result differs from the analytic
aggregates (a race in the slice pipeline shows up as an occasional wrong step, not as a wrong final step).
usage: python tools/stress_cpra.py [--world 1 --transport rccl|loopback --slices 8 --steps 40 --inner N --outer N --option name=value ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--transport", default="rccl")
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--unique", action="store_true", help="HJGPU_FLAG_UNIQUE (same result here: the build keys are unique)")
    ap.add_argument("--forensics", action="store_true", help="communicator option debug_forensics: every stage of every step leaves "
                    "checksums (option audit); a wrong step prints the stage whose output lacked the tuples")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--ctx-option", action="append", default=[], help="hjgpu_set_option on every rank's join context")
    a = ap.parse_args()
    import ctypes
    import torch
    torch.cuda.init()
    import hash_join_codes_knl_amd as H
    from hash_join_codes_knl_amd import api
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    lib = api.load_library()
    knobs = {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "GPU_MAX_HW", "HJGPU_LIBRARY", "AMD_SERIALIZE", "HIP_LAUNCH_BLOCKING"))}
    print("library %s, kernel hash %s, env %s" % (os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), H.kernel_hash(), knobs), flush=True)
    has_dbg = hasattr(lib, "hjgpu_debug_scratch")
    if has_dbg:
        lib.hjgpu_debug_scratch.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        lib.hjgpu_debug_scratch(None, 1)
    comm = H.HjComm.local(a.world, [0] * a.world, H.TRANSPORT_RCCL if a.transport == "rccl" else H.TRANSPORT_LOOPBACK)
    for o in a.option:
        n, v = o.split("=")
        comm.set_option(n, int(v))
    for o in a.ctx_option:
        n, v = o.split("=")
        for ctx in comm.ctx:
            ctx.set_option(n, v)
    fi, fo = 0x2545F491, 0x9E3779B1
    G = a.world
    cols, shards, expect = [], [], [0, 0, 0, 0]
    for g in range(G):
        ctx = comm.ctx[g]
        ri, ro = a.inner // G, a.outer // G
        c = [ctx.column(ri), ctx.column(ri), ctx.column(ro), ctx.column(ro)]
        ctx.generate_range(1, ri * G, ro * G, g * ri, ri, g * ro, ro, fi, fo, *c)
        sums = ctx.column_sums(c[2], ro, fo, fi)
        expect = [expect[0] + ro] + [(x + y) & ((1 << 64) - 1) for x, y in zip(expect[1:], sums)]
        cols += c
        shards.append((c[0], c[1], ri, c[2], c[3], ro))
    if a.forensics:
        comm.set_option("debug_forensics", 1)
    M = (1 << 64) - 1

    def report(step):
        """names every stage of the step whose checksum differs from what the stage read (workload: selectivity 1, unique build keys)"""
        found = []
        for rank, parts, joins in comm.forensics():
            for i, rec in enumerate(parts):          # partitioning calls: build side, then the probe slices
                what = "rank %d partitioning call %d (%s)" % (rank, i, "build side" if i == 0 else "probe slice %d" % (i - 1))
                if rec[1][0] or rec[1][1:] != rec[0][1:]:
                    found.append("%s: pass-1 output: %d misplaced, %+d tuples, key sum %+d vs input" % (what, rec[1][0], rec[1][3] - rec[0][3], (rec[1][1] - rec[0][1] + (1 << 63)) % (1 << 64) - (1 << 63)))
            build_final = None
            for i, rec in enumerate(joins):
                kind = rec[7][1]
                what = "rank %d join call %d (%s)" % (rank, i, {1: "build", 2: "probe"}.get(kind, "kind %d" % kind))
                side = 3 if kind == 1 else 0
                if rec[side + 2][0] or rec[side + 2][1:] != rec[side][1:]:
                    found.append("%s: final partitions of the %s side: %d misplaced, %+d tuples vs what the call read" % (what, "build" if kind == 1 else "probe", rec[side + 2][0], rec[side + 2][3] - rec[side][3]))
                if kind == 1:
                    build_final = rec[5]
                elif build_final is not None and rec[5] != build_final:
                    found.append("%s: the prepared build side changed since its build: %s -> %s" % (what, build_final, rec[5]))
                if kind == 2 and (rec[6][0] != rec[0][3] or rec[6][1] != rec[0][1] or rec[6][2] != rec[0][2]):
                    found.append("%s: join result count %+d, key sum %+d vs the batch it read (partitions right: %s)" % (what, rec[6][0] - rec[0][3], (rec[6][1] - rec[0][1] + (1 << 63)) % (1 << 64) - (1 << 63), not rec[2][0] and rec[2][1:] == rec[0][1:]))
            if G == 1 and len(parts) == len(joins):
                for i, (p_, j_) in enumerate(zip(parts, joins)):
                    side = 3 if j_[7][1] == 1 else 0
                    if p_[1][1:] != j_[side][1:]:
                        found.append("rank %d call %d: the join read %+d tuples, key sum %+d vs what the partitioning left" % (rank, i, j_[side][3] - p_[1][3], (j_[side][1] - p_[1][1] + (1 << 63)) % (1 << 64) - (1 << 63)))
        print("step %d forensics: %s" % (step, "; ".join(found) if found else "every stage's checksums agree"), flush=True)

    bad = 0
    for s in range(a.steps):
        if s and s % 1000 == 0:
            print("... %d steps, %d wrong so far" % (s, bad), flush=True)
        got, st = comm.cpra_multi(shards, H.PhjParams(flags=H.FLAG_UNIQUE) if a.unique else None, a.slices)
        if list(got) != expect:
            bad += 1
            print("step %d WRONG: count %+d, sums %s" % (s, got[0] - expect[0], ["%+d" % ((x - y + (1 << 63)) % (1 << 64) - (1 << 63)) for x, y in zip(got[1:], expect[1:])]), flush=True)
            if a.forensics:
                report(s)
        elif a.forensics and s == 0:
            report(s)
    print("%s world %d slices %d options %s: %d of %d steps wrong, last %.2f ms" % (a.transport, G, a.slices, a.option + a.ctx_option + (["unique"] if a.unique else []), bad, a.steps, st["ms_wall"]), flush=True)
    if has_dbg:
        # HJ_SCRATCH_EXPERIMENT variants 2-4: values that came back from the private segment, compared in the kernel
        d = (ctypes.c_uint64 * 40)()
        lib.hjgpu_debug_scratch(d, 0)
        print("private segment: %d values re-read and compared in the kernel, %d MISMATCHES" % (d[1], d[0]), flush=True)
        if d[5]:
            print("pass 1, time between two tiles of a workgroup: longest %.1f us; %d of %d gaps > 100 us, %d > 1 ms" % (d[2] / 100.0, d[4], d[5], d[3]), flush=True)
        for i in range(min(int(d[0]), 8)):
            print("    block %d thread %d slot %d: expected %016x got %016x" % (d[8 + 4 * i] >> 32, d[8 + 4 * i] & 0xFFFFFFFF, d[9 + 4 * i], d[10 + 4 * i], d[11 + 4 * i]), flush=True)
    comm.close()
    return 1 if bad else 0       # evidence scripts stop on a wrong step (round 4's collected a file with WRONG in it silently)


if __name__ == "__main__":
    sys.exit(main())

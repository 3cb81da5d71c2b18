#!/usr/bin/env python3
"""Repeats hjgpu_cpra_multi (or, --algo phj / npj, hjgpu_phj_multi / hjgpu_npj_multi) on one communicator and reports every step
whose result differs from the analytic aggregates (a race in a pipeline shows up as an occasional wrong step, not as a wrong final step).
usage: python tools/stress_cpra.py [--algo cpra|phj|npj --world 1 --transport rccl|loopback --slices 8 --steps 40 --inner N --outer N --option name=value ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="cpra", choices=["cpra", "phj", "npj"], help="phj / npj: the build side replicated from rank 0, the probe side sharded")
    ap.add_argument("--rows", action="store_true", help="materialise the result (hjgpu_*_multi_rows: per-rank dense rows) and check the ROWS every step: their number and the "
                    "sums of the three result columns, read back by a kernel of its own (a lost row store or close_gaps move does not change the join's aggregates)")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--transport", default="rccl")
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--inner", type=int, default=64_000_000)
    ap.add_argument("--outer", type=int, default=1_000_000_000)
    ap.add_argument("--unique", action="store_true", help="HJGPU_FLAG_UNIQUE (same result here: the build keys are unique)")
    ap.add_argument("--forensics", action="store_true", help="communicator option debug_forensics: every stage of every step leaves "
                    "checksums (option audit); a wrong step prints the stage whose output lacked the tuples")
    ap.add_argument("--freeze", action="store_true", help="debug_forensics = 2: every slice's records are read as soon as the slice is done and a wrong stage is looked "
                    "at again at once - by a fresh kernel and from a hipMemcpy on the host - while its buffers are intact (any number of slices)")
    ap.add_argument("--recheck-first", action="store_true", help="with --forensics: step 0's report includes the second look (hjgpu_comm_recheck) although the step is right")
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
    print("library %s, library hash %s, kernel hash %s, env %s" % (os.path.basename(os.environ.get("HJGPU_LIBRARY", "libhjgpu.so")), H.library_hash(), H.kernel_hash(), knobs), flush=True)
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
        if a.algo != "cpra":
            # replicated build side: all of it on rank 0 (the root), nothing elsewhere; the probe side in G shards
            sk, sv = ctx.column(ro), ctx.column(ro)
            rk = rv = None
            if g == 0:
                rk, rv = ctx.column(ri * G), ctx.column(ri * G)
                ctx.generate_range(1, ri * G, ro * G, 0, ri * G, 0, ro, fi, fo, rk, rv, sk, sv)
            else:
                scratch = [ctx.column(16), ctx.column(16)]
                ctx.generate_range(1, ri * G, ro * G, 0, 16, g * ro, ro, fi, fo, scratch[0], scratch[1], sk, sv)
            sums = ctx.column_sums(sk, ro, fo, fi)
            expect = [expect[0] + ro] + [(x + y) & ((1 << 64) - 1) for x, y in zip(expect[1:], sums)]
            cols += [sk, sv]
            shards.append((rk, rv, ri * G, sk, sv, ro))
            continue
        c = [ctx.column(ri), ctx.column(ri), ctx.column(ro), ctx.column(ro)]
        ctx.generate_range(1, ri * G, ro * G, g * ri, ri, g * ro, ro, fi, fo, *c)
        sums = ctx.column_sums(c[2], ro, fo, fi)
        expect = [expect[0] + ro] + [(x + y) & ((1 << 64) - 1) for x, y in zip(expect[1:], sums)]
        cols += c
        shards.append((c[0], c[1], ri, c[2], c[3], ro))
    if a.forensics or a.freeze:
        comm.set_option("debug_forensics", 2 if a.freeze else 1)
    M = (1 << 64) - 1

    def report(step, second_look=True):
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
        # second look with the device quiet (hjgpu_comm_recheck): the LAST partitioning call and the LAST join of every rank - with
        # --slices 1 that is the whole probe side - by a fresh kernel and on the host from a hipMemcpy of the partitions
        if a.freeze:
            ev = comm.frozen()
            if not ev:
                print("    freeze: no stage was found wrong while its buffers were intact", flush=True)
            for rank, which, sl, rec, checks in ev:
                kind = rec[7][1]
                res = rec[6]
                print("    freeze: rank %d %s context, slice %d, call kind %d: record %s" % (rank, "join" if which else "partitioning", sl, kind, rec[:7]), flush=True)
                for stage, fresh, host in checks:
                    want = rec[3 if stage in (4, 5) else 0]
                    if stage == 5 and kind == 2:
                        want = None                            # (the build call's input is not in this record: compare the three views only)
                    ok = (lambda x: x[0] == 0 and x[1:] == want[1:]) if want else (lambda x: x == host)
                    if ok(rec[stage]):
                        verdict = "right on the call's stream" + ("" if ok(fresh) and ok(host) else ", WRONG now")
                    elif not ok(host):
                        verdict = "MEMORY IS WRONG (hipMemcpy to the host%s): stores LOST" % (", the fresh kernel alike" if fresh == host else "; the fresh kernel sees %s" % fresh)
                    elif ok(fresh):
                        verdict = "memory is RIGHT now (host copy and fresh kernel): the call's own check read STALE data"
                    else:
                        verdict = "host copy right, fresh kernel wrong"
                    print("        stage %d: on the call's stream %s | fresh kernel, device quiet %s | host copy %s -> %s" % (stage, rec[stage], fresh, host, verdict), flush=True)
                if kind == 2 and (res[0] != rec[0][3] or res[1] != rec[0][1]):
                    print("        the join's result: count %+d vs the batch it read" % (res[0] - rec[0][3]), flush=True)
            return
        if not second_look:
            return
        ranks = {r: (p_, j_) for r, p_, j_ in comm.forensics()}
        for rank, which, checks in comm.recheck():
            recs = ranks[rank][which]
            if not recs:
                continue
            first = recs[-1]                                   # the record of the context's last call: what the checks saw on the call's stream
            for stage, fresh, host in checks:
                want = first[3 if stage in (4, 5) else 0]      # what the call read: stage 0 (probe side / a partitioning call's input), 3 (build side)
                if stage in (4, 5) and not any(want):          # a probe of a prepared build side: what the BUILD call read
                    want = recs[0][3]
                ok = lambda x: x[0] == 0 and x[1:] == want[1:]
                if ok(first[stage]):
                    verdict = "right all along" if ok(fresh) and ok(host) else "RIGHT on the call's stream, wrong now"
                elif not ok(host):
                    verdict = "MEMORY IS WRONG (hipMemcpy to the host%s): the stores are LOST" % (", fresh kernel alike" if fresh == host else "; the fresh kernel sees something else again")
                elif ok(fresh):
                    verdict = "memory is RIGHT now (host copy and fresh kernel): the check on the call's stream read STALE data"
                else:
                    verdict = "host copy right, fresh kernel wrong"
                print("    rank %d %s context, stage %d: on the call's stream %s | fresh kernel, device quiet %s | host copy %s | input %s -> %s"
                      % (rank, "join" if which else "partitioning", stage, first[stage], fresh, host, want, verdict), flush=True)

    outs = None
    if a.rows:
        # every rank's result columns: its share of the rows (selectivity 1: one row per probe tuple) plus the open blocks' slack
        outs, block = [], 4096
        for g in range(G):
            ctx = comm.ctx[g]
            cap = ctx.output_capacity(0 if a.algo == "npj" else 1, a.outer // G, 2 * (a.outer // G) if a.algo == "cpra" else a.outer // G, block)
            outs.append((ctx.column(cap, placed=True), ctx.column(cap, placed=True), ctx.column(cap, placed=True), cap, block))
    bad = 0
    for s in range(a.steps):
        if s and s % 1000 == 0:
            print("... %d steps, %d wrong so far" % (s, bad), flush=True)
        if a.rows:
            pp = H.PhjParams(flags=H.FLAG_UNIQUE) if a.unique else None
            if a.algo == "cpra":
                got, st, counts = comm.cpra_multi_rows(shards, outs, pp, a.slices)
            elif a.algo == "phj":
                got, st, counts = comm.phj_multi_rows(shards, outs, 0, pp)
            else:
                got, st, counts = comm.npj_multi_rows(shards, outs, 0, H.NpjParams(flags=H.FLAG_UNIQUE) if a.unique else None)
            # the rows themselves: every rank's dense prefix summed by a kernel of its own, column by column
            rows = [sum(counts), 0, 0, 0]
            for g in range(G):
                for c in range(3):
                    rows[c + 1] = (rows[c + 1] + comm.ctx[g].column_sums(outs[g][c], counts[g], 1, 1)[0]) & ((1 << 64) - 1)
            if rows != expect:
                bad += 1
                print("step %d WRONG ROWS: count %+d, column sums %s" % (s, rows[0] - expect[0], ["%+d" % ((x - y + (1 << 63)) % (1 << 64) - (1 << 63)) for x, y in zip(rows[1:], expect[1:])]), flush=True)
        elif a.algo == "cpra":
            got, st = comm.cpra_multi(shards, H.PhjParams(flags=H.FLAG_UNIQUE) if a.unique else None, a.slices)
        elif a.algo == "phj":
            got, st = comm.phj_multi(shards, 0, H.PhjParams(flags=H.FLAG_UNIQUE) if a.unique else None)
        else:
            got, st = comm.npj_multi(shards, 0, H.NpjParams(flags=H.FLAG_UNIQUE) if a.unique else None)
        if list(got) != expect:
            bad += 1
            print("step %d WRONG: count %+d, sums %s" % (s, got[0] - expect[0], ["%+d" % ((x - y + (1 << 63)) % (1 << 64) - (1 << 63)) for x, y in zip(got[1:], expect[1:])]), flush=True)
            if a.forensics or a.freeze:
                report(s)
        elif (a.forensics or a.freeze) and s == 0:
            report(s, a.recheck_first)
    print("%s%s %s world %d slices %d options %s: %d of %d steps wrong, last %.2f ms" % (a.algo, " with rows (checked on their own)" if a.rows else "", a.transport, G, a.slices, a.option + a.ctx_option + (["unique"] if a.unique else []), bad, a.steps, st["ms_wall"]), flush=True)
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

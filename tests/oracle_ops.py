"""Test infrastructure: the data-path operators that distributed.py expects, implemented with the CPU oracle,
so that the multi-GPU host logic runs over gloo without a GPU (test_distributed_gloo.py)."""
import numpy as np


class OracleOps:
    """The operator interface of hash_join_codes_knl_amd.distributed (join, partition) over the CPU oracle."""

    def __init__(self, oracle, torch):
        self.O, self.torch = oracle, torch

    def _np(self, t):
        return t.numpy().view(np.uint32)

    def join(self, rk, rv, sk, sv):
        if rk.numel() == 0 or sk.numel() == 0:
            return (0, 0, 0, 0)
        return self.O.join_definition(self._np(rk), self._np(rv), self._np(sk), self._np(sv))

    def partition(self, keys, vals, factor, fanout):
        counts, ko, vo = self.O.partition(self._np(keys), self._np(vals), factor, fanout)
        off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
        return (self.torch.from_numpy(ko.view(np.int32)), self.torch.from_numpy(vo.view(np.int32)),
                [int(x) for x in off])


class PreparedOracleOps(OracleOps):
    """The same with the prepared-build pair (GpuOps.prepare_build / probe): the sliced exchange then
    prepares the received build side once and probes one batch per slice."""
    supports_prepared_build = True

    def __init__(self, oracle, torch):
        super().__init__(oracle, torch)
        self.builds, self.probes = 0, 0

    def prepare_build(self, rk, rv, max_outer):
        self.built = (rk, rv, max_outer)
        self.builds += 1

    def probe(self, sk, sv):
        self.probes += 1
        rk, rv, _ = self.built
        return self.join(rk, rv, sk, sv)

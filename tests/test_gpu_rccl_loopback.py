"""-m gpu: libcssm_pf's OWN series loop over RCCL (cssm_pf_shard_series_rccl: the library enqueues kernels and collectives itself, what a node
falls back to if the peer-written windows are refused) at world 2, 4 and 8 on ONE GPU.

RCCL admits one rank per device and process, so that loop had only ever run at world 1.  Here every rank is a
THREAD of this process with a shard, a stream and a ShardedFilter of its own, and the nccl* entry points the library resolves at run time
come from tests/cpp/rccl_loopback.cpp (CSSM_RCCL_LIB): the same send / receive buffers, counts, displacements and call order as on a node
-- ncclAllToAll, the trimmed ncclAllToAllv (whole segments between adjacent ranks, 12 header words between the others) and the
ncclAllGather ahead of an observation whose level comes from the global max -- the data moved by device-to-device copies behind events.
Every rank must return the single-rank oracle's bits.  What one GPU cannot show is the transport.

The test runs in a child process: the library resolves its RCCL once per process, and the rest of the suite must keep the real one.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import ctypes as C, os, sys, threading
import numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from composablestatespacemodels_amd import _abi
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
from oracle import oracle


class ThreadComm:
    """One shard per THREAD: the control plane of DistComm (agreements, the host-driven exchange of a resumed observation) through a
    threading.Barrier, the per-observation collectives through the library's own communicator (native_comm)."""
    peer = False

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.sh = rank, world, shared
        self._native = None

    def barrier(self):
        self.sh["bar"].wait()

    def native_comm(self):
        if self._native is None:
            lib = _abi.load_library()
            if self.rank == 0:
                ident = C.create_string_buffer(128)
                _abi.check(lib.cssm_rccl_unique_id(ident))
                self.sh["id"] = ident.raw
            self.barrier()
            h = C.c_void_p()
            _abi.check(lib.cssm_rccl_comm_create(C.create_string_buffer(self.sh["id"], 128), self.world, self.rank, 0, C.byref(h)))
            self._native = h
        return self._native

    def agree_max(self, values):
        self.sh["vals"][self.rank] = int(values[0])
        self.barrier()
        m = max(self.sh["vals"])
        self.barrier()
        return m

    def _swap(self, ins, fn):
        torch.cuda.current_stream().synchronize()
        self.sh["ins"][self.rank] = ins[0]
        self.barrier()
        fn(self.sh["ins"])
        torch.cuda.current_stream().synchronize()
        self.barrier()

    def all_to_all_equal(self, outs, ins):
        seg = ins[0].numel() // self.world
        self._swap(ins, lambda peers: [outs[0][q * seg:(q + 1) * seg].copy_(peers[q][self.rank * seg:(self.rank + 1) * seg]) for q in range(self.world)])

    def all_gather(self, outs, ins):
        n = ins[0].numel()
        self._swap(ins, lambda peers: [outs[0][q * n:(q + 1) * n].copy_(peers[q]) for q in range(self.world)])

    # (the host-read exact exchange, where a plan ends on it: slots owned by particles of non-adjacent ranks)
    def all_to_all_counts(self, outs, ins):
        self._swap(ins, lambda peers: [outs[0][r:r + 1].copy_(peers[r][self.rank:self.rank + 1]) for r in range(self.world)])

    def all_to_all_v(self, outs, ins, out_splits, in_splits):
        self.sh["splits"][self.rank] = [int(v) for v in in_splits[0]]

        def move(peers):
            pos = 0
            for r in range(self.world):
                sp = self.sh["splits"][r]
                off, n = sum(sp[:self.rank]), sp[self.rank]
                assert n == int(out_splits[0][r])
                if n:
                    outs[0][pos:pos + n].copy_(peers[r][off:off + n])
                pos += n
        self._swap(ins, move)

    def all_reduce_sum(self, tensors):
        def add(peers):
            total = peers[0].clone()
            for x in peers[1:]:
                total += x
            self.sh["sum"][self.rank] = total
        self._swap(tensors, add)
        tensors[0].copy_(self.sh["sum"][self.rank])

    def combine_rows(self, arrays):
        self.sh["rows"][self.rank] = np.ascontiguousarray(arrays[0]).view(np.uint64)
        self.barrier()
        out = np.zeros(arrays[0].shape, dtype=np.uint64)
        for a in self.sh["rows"]:
            out |= a.reshape(out.shape)
        self.barrier()
        return out.view(np.float64)

    def close(self):
        if self._native:
            _abi.load_library().cssm_rccl_comm_destroy(self._native)


def run(world, name, n, T, prec, mode, outlier, tiny_cap):
    model = getattr(cases, name)()
    t, y, has = cases.event_times(T, horizon=0.1 * T) if prec else cases.poisson_counts(T, missing=0.2)
    y = y.copy()
    if outlier:
        y[T // 2] = 70.0; has[T // 2] = 1
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
    ol, _, oess, _ = o.filter(t, y, has)
    shared = {"bar": threading.Barrier(world), "vals": [0] * world, "ins": [None] * world, "splits": [None] * world, "sum": [None] * world, "rows": [None] * world}
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                shard = GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec)
                f = ShardedFilter([shard], ThreadComm(r, world, shared))
                f.SINGLE_MODE = mode
                if tiny_cap:
                    f.MIN_CAP, f.CAP_SQRT = 64, 0.0          # a capacity the series must miss: resumed, the exchange redone host-driven
                if outlier:       # (on the collectives an outlying observation voids the plan: the series is repeated from its start, every
                    ll, ess = f.ll_filter(t, y, has, lgcp=bool(prec))   #  level from the all-gathered max -- which a continued part could not be)
                else:
                    a = T // 2
                    f.ll_filter(t[:a], y[:a], has[:a], lgcp=bool(prec))
                    ll, ess = f.ll_filter_more(t[a:], y[a:], has[a:], lgcp=bool(prec))
                out[r] = (ll, ess, f.last_native, f.last_resumes, f.last_all_to_all, f.last_from_max)
                torch.cuda.current_stream().synchronize()
                f.comm.close(); shard.close()
        except BaseException as e:   # noqa: BLE001 -- a rank that dies must not leave its peers in a barrier
            errs.append((r, repr(e)))
            shared["bar"].abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join(300) for x in th]
    assert not errs, errs
    assert all(x is not None for x in out), "a rank did not finish"
    for r, (ll, ess, native, resumes, how, from_max) in enumerate(out):
        assert native, f"rank {r}: the series did not run on the library's own collectives"
        assert from_max == outlier
        assert ll == ol and ess == int(oess[-1]), (r, ll, ol, ess, int(oess[-1]))
    print(f"world {world} {name} N={n} T={T} mode {mode}: every rank == the oracle (ll {ol!r}); {out[0][4]}; resumes {out[0][3]}, from max {out[0][5]}")


for case in CASES:
    run(*case)
print("LOOPBACK OK")
'''

CASES = [
    # world, model, N, T, LGCP precision, exchange mode (1 ncclAllToAll, 3 trimmed ncclAllToAllv at any world), outlying observation, tiny capacity
    (2, "c2_model", 40000, 12, 0, 1, False, False),
    (4, "c2_model", 70001, 12, 0, 3, False, False),
    (2, "c2_model", 50000, 10, 0, 1, True, False),      # an outlier: the series repeated with every level from the all-gathered max (ncclAllGather ahead of every
                                                        # exchange; world 2: the two ranks are adjacent, boundary blocks cover whatever the weights do)
    (8, "c1_model", 80000, 10, 0, 1, False, True),      # capacity misses: resumed, redone through the host-driven exchange, the series goes on
    (4, "c4_model", 60000, 8, 2, 3, False, False),      # LGCP: the first event's level from the all-gathered max, then predicted levels
]


def test_library_series_loop_over_rccl_at_world_2_4_8_in_process(tmp_path):
    so = tmp_path / "librccl_loopback.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-o", str(so),
                           os.path.join(HERE, "cpp", "rccl_loopback.cpp")], stderr=subprocess.DEVNULL)
    env = dict(os.environ, CSSM_RCCL_LIB=str(so), CSSM_SHARD_PEER="0")
    src = f"ROOT = {ROOT!r}\nCASES = {CASES!r}\n" + CHILD
    r = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and "LOOPBACK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]

#!/usr/bin/env python3
"""When the eager rows' flag leaves: clock stamps of the pack launch (headers + eager rows) of a MIDDLE shard of `world` in-process shards --
per destination, the block that set the ROWS flag against the header block's "header + flag out".  The last observation is driven stage by
stage so that the middle shard's pack launch is the last one before the stamps are read.  Diagnostic build as tools/archive/exchange_stamps.py.
Run:   CSSM_PF_LIB=.../build_stamps/libcssm_pf_stamps.so python tools/archive/pack_stamps_local.py [world=8] [particles per shard=1048576]"""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases  # noqa: E402
from composablestatespacemodels_amd import _abi  # noqa: E402
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter  # noqa: E402
from local_comm import LocalCommPeer  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
lib = _abi.load_library()
lib.cssm_pf_debug_spec_stamps.restype = C.c_int
lib.cssm_pf_debug_spec_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
model = cases.c2_model(); t, y, has = cases.poisson_counts(24, missing=0.0)
shards = [GpuShard(model, world * per, r, world, cases.SEED, 0) for r in range(world)]
f = ShardedFilter(shards, LocalCommPeer(world))
f.ll_filter(t[:20], y[:20], has[:20])
cap = f.last_cap
mid = world // 2
for s in shards:
    s.begin_more(t[20:], y[20:], has[20:])
for k in range(len(t) - 20):
    for s in shards:
        s.propagate_at(k, with_sums=False)
    order = [s for s in shards if s.rank != mid] + [shards[mid]]     # the middle shard's pack launch last
    for s in order:
        s.pack_peer(cap)
    if k == len(t) - 21:
        import torch
        torch.cuda.synchronize()
        out = np.zeros(2048 * 8, dtype=np.uint64)
        assert lib.cssm_pf_debug_spec_stamps(shards[mid]._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size) == 0
        st = out.reshape(2048, 8).astype(np.int64)
    for s in shards:
        s.pack_rows_peer(cap)
    for s in shards:
        s.adopt_peer(cap)
gx = -(-min(cap, per) // 1024) + 2
# (the standalone pack launch is a (gx, world) grid and the stamps are indexed by blockIdx.x: row bx holds the stamps of the destination
#  whose block bx stamped last -- only the two neighbours' row blocks stamp at all)
hb = st[gx - 2]
t0 = int(hb[4])
print(f"world {world}, shard {mid} of {per} particles, capacity {cap} rows = {gx - 2} row blocks per destination; us after a header block's entry")
print(f"  a header block: first instruction {(hb[0] - t0) * 0.01:5.2f}, entered {0.0:5.2f}, sums in LDS {(hb[1] - t0) * 0.01:5.2f}, header words stored {(hb[2] - t0) * 0.01:5.2f}, header + flag out {(hb[5] - t0) * 0.01:5.2f}")
for bx in range(gx - 2):
    r = st[bx]
    if r[0] >= t0 - 100000:
        f = lambda k: f"{(r[k] - t0) * 0.01:6.2f}" if r[k] >= t0 - 100000 else "     -"
        print(f"  row block {bx:2d}: entered {f(0)}, prefix known {f(1)}, eager rows stored {f(2)}, behind fence + barrier {f(7)}, ROWS flag out {f(3)}")
for s_ in shards:
    s_.close()

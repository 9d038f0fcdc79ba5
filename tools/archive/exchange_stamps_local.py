#!/usr/bin/env python3
"""Clock stamps of the adopt launch (k_offspring_expand_spec, the peer-written exchange in stages) of the LAST of `world` in-process shards:
where a shard WITH a neighbour spends its time against world 1.  Diagnostic build as in tools/archive/exchange_stamps.py (-DCSSM_OFF_STAMPS).
Run:   CSSM_PF_LIB=.../build_stamps/libcssm_pf_stamps.so python tools/archive/exchange_stamps_local.py [world=2] [particles per shard=1048576]
Stamps (100 MHz): 0 entry, 7 all headers seen, 1 headers in LDS + level checked, 2 own ancestors written (offspring blocks) / 3 rows expanded
(the 64 expansion blocks that lead the grid)."""
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

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
lib = _abi.load_library()
lib.cssm_pf_debug_spec_stamps.restype = C.c_int
lib.cssm_pf_debug_spec_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
model = cases.c2_model(); t, y, has = cases.poisson_counts(40)
XB = 64
for w in (1, world):
    shards = [GpuShard(model, w * per, r, w, cases.SEED, 0) for r in range(w)]
    f = ShardedFilter(shards, LocalCommPeer(w))
    f.ll_filter(t, y, has)
    assert f.last_peer, "the peer-written exchange did not run"
    out = np.zeros(2048 * 8, dtype=np.uint64)
    assert lib.cssm_pf_debug_spec_stamps(shards[-1]._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size) == 0
    s = out.reshape(2048, 8).astype(np.int64)
    live = s[:, 0] > 0
    t0 = s[live][:, 0].min()
    print(f"world {w}, shard {w - 1}, {per} particles per shard, cap {f.last_cap}: {int(live[XB:].sum())} offspring blocks, {int(live[:XB].sum())} expansion blocks")
    for lo, hi, what, keys in ((XB, 2048, "offspring", ((0, "entry"), (7, "all headers seen"), (1, "headers in LDS, level ok"), (2, "own ancestors written"))),
                               (0, XB, "expansion", ((0, "entry"), (3, "rows expanded")))):
        blk = s[lo:hi][live[lo:hi]]
        for k, name in keys:
            v = blk[:, k]
            v = (v[v > 0] - t0) * 0.01
            if len(v):
                print("  %-10s %-28s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f  (%d blocks)" % (what, name, v.min(), np.median(v), np.percentile(v, 90), v.max(), len(v)))
    late = np.argsort(-s[XB:, 2])[:6]       # the offspring blocks that wrote their ancestors last
    print("  last offspring blocks (index: entry / headers seen / headers in LDS / ancestors written): " +
          "; ".join("%d: %.2f / %.2f / %.2f / %.2f" % ((b,) + tuple((s[XB + b, k] - t0) * 0.01 for k in (0, 7, 1, 2))) for b in late))
    for sh in shards:
        sh.close()

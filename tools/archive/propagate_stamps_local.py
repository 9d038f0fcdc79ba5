#!/usr/bin/env python3
"""Clock stamps of k_propagate_shard of the LAST of `world` in-process shards (a shard that gathers rows its neighbour wrote) against
world 1: which blocks finish last.  Diagnostic build as in tools/archive/propagate_stamps.py (-DCSSM_PROP_STAMPS -DCSSM_PROP_D=3).
Run:   CSSM_PF_LIB=.../build_stamps/libcssm_pf_pstamps.so python tools/archive/propagate_stamps_local.py [world=2] [particles per shard=1048576]"""
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
fn = lib.cssm_prop_debug_stamps_d3
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
model = cases.c2_model(); t, y, has = cases.poisson_counts(40)
names = {0: "entry", 1: "behind the table's barrier", 3: "first tile's rows landed", 4: "first tile computed", 6: "last tile computed", 7: "block done"}
for w in (1, world):
    shards = [GpuShard(model, w * per, r, w, cases.SEED, 0) for r in range(w)]
    f = ShardedFilter(shards, LocalCommPeer(w))
    f.ll_filter(t, y, has)
    nb = (per + 1023) // 1024
    out = np.zeros(nb * 8, dtype=np.uint64)
    assert fn(out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size) == 0
    s = out.reshape(nb, 8).astype(np.int64)
    t0 = s[:, 0].min()
    print(f"world {w}, shard {w - 1}, {per} particles per shard, {nb} blocks (us after the launch's earliest block entry)")
    for k, name in names.items():
        v = (s[:, k] - t0) * 0.01
        print("  %-28s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    late = np.argsort(-s[:, 7])[:6]
    print("  last blocks (index: entry / rows landed / first tile computed / done): " +
          "; ".join("%d: %.2f / %.2f / %.2f / %.2f" % ((b,) + tuple((s[b, k] - t0) * 0.01 for k in (0, 3, 4, 7))) for b in late))
    for sh in shards:
        sh.close()

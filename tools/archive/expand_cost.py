"""What the expansion of the neighbours' rows costs the resampling launch: `world` shards of ONE process on this GPU (peer-written exchange in
stages: pack / pack rows / adopt, tests/local_comm.py), every launch bracketed by HIP events; the adopt launch (k_offspring_expand_spec) of a
middle shard (two neighbours) against the same launch at world 1 (no neighbour, no rows).  Bench workload (C2).
usage: expand_cost.py [world=4] [particles per shard=1048576] [observations=200] [lgcp]   (lgcp: BASELINE configs[3] instead of the bench workload)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
from local_comm import LocalCommPeer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lgcp = len(sys.argv) > 4 and sys.argv[4] == "lgcp"
if lgcp:
    model = cases.c4_model(); t, y, has = cases.event_times(T + 20, horizon=0.1 * (T + 20))
else:
    model = cases.c2_model(); t, y, has = cases.poisson_counts(T + 20, missing=0.05)
for w in (1, world):
    shards = [GpuShard(model, w * per, r, w, cases.SEED, 0, lgcp_precision=2 if lgcp else 0) for r in range(w)]
    f = ShardedFilter(shards, LocalCommPeer(w))
    f.ll_filter(t[:20], y[:20], has[:20], lgcp=lgcp)
    for s in shards:
        s.profile(True)
    f.ll_filter_more(t[20:], y[20:], has[20:], lgcp=lgcp)
    for r, s in enumerate(shards):
        p = s.profile_read()
        print(f"world {w} shard {r}: " + ", ".join(f"{k} {v[0] / v[1] * 1e3:.2f} us x {v[1]}" for k, v in p.items() if v[1]))
    assert f.last_peer
    for s in shards:
        s.close()

import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, cases
from composablestatespacemodels_amd.sharded import GpuShard, LocalComm, ShardedFilter
t, y, has = cases.poisson_counts(300, missing=0.05)
model = cases.c2_model()
n, world = 1 << 20, 8
for single in ("1", "0"):
    os.environ["CSSM_SHARD_SINGLE"] = single
    shards = [GpuShard(model, n, r, world, 77, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    orig_status = {}
    allbits = []
    for s in shards:
        st = s.status
        def wrapped(T, st=st):
            r = st(T); allbits.append((r[2], int(r[3].max()))); return r
        s.status = wrapped
    ll, ess = f.ll_filter(t[:120], y[:120], has[:120])
    print("single", single, "attempts", f.last_attempts, "cap", f.last_cap, "(bits, max need) per shard per attempt:", allbits)
    for s in shards: s.close()

"""How many boundary rows does the bench workload need per observation at world = W (local shards, one GPU)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, cases
from composablestatespacemodels_amd.sharded import GpuShard, LocalComm, ShardedFilter
MODE = sys.argv[1] if len(sys.argv) > 1 else "need"
os.environ["CSSM_SHARD_SINGLE"] = "0" if MODE == "need" else "1"   # the two-collective exchange records the rows every step needed
t, y, has = cases.poisson_counts(int(os.environ.get("PROBE_T", "500")))
model = cases.c2_model()
for world in (2, 4, 8):
    n = (1 << 20) * world
    shards = [GpuShard(model, n, r, world, 20260101, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    if MODE == "need":
        f.MIN_CAP = 1 << 17                    # generous: nothing overflows, we only want the need statistics
    needs = []
    for s in shards:
        st = s.status
        def wrapped(T, st=st):
            r = st(T); needs.append(r[3].copy()); return r
        s.status = wrapped
    import time, torch
    f.ll_filter(t[:12], y[:12], has[:12])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ll, ess = f.ll_filter(t, y, has)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print(f"   {wall / len(t) / world * 1e6:.1f} us per observation and shard (all shards on one GPU, collectives = copies)")
    needs = [a for a in needs if a.shape == (len(t),)]
    need = np.max(np.stack(needs), axis=0)
    if MODE != "need":
        from composablestatespacemodels_amd.filter import NativePf
        pf = NativePf(model, n, 20260101); l1 = pf.run(t, y, has)[0]; pf.close()
        print(f"   resumes {f.last_resumes}; ll sharded {ll!r} single-GPU {l1!r} {'EQUAL' if ll == l1 else 'DIFFERENT'}")
    print(f"world {world} N {n}: attempts {f.last_attempts} cap {f.last_cap}; rows needed per observation: max {need.max()} p99 {int(np.percentile(need, 99))} "
          f"median {int(np.median(need))}; first 6 {need[:6].tolist()}; sqrt(N) = {int(n ** 0.5)}", flush=True)
    for s in shards: s.close()

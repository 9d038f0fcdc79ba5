"""Run-to-run determinism soak (run on the GPU box): the same long series several times, single GPU (separate and fused
sums, several dimensions) and 4 local shards with the single-collective exchange; every repetition must give the same
bits (ll, ess trace, ancestors, particles)."""
import hashlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from local_comm import LocalComm

def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]

bad = 0
t, y, has = cases.poisson_counts(300, missing=0.05)
for name, n, fused in (("c2_model", (1 << 22) + 333, 0), ("c2_model", 1 << 22, 1), ("c1_model", 1 << 23, 0), ("c3_model", 1 << 21, 0), ("max_dim_model", 1 << 19, 1),
                       ("c2_model", 100_000, 1), ("c2_model", (1 << 20) - 5, 1), ("c3_model", (1 << 21) + 7, 1), ("max_dim_model", (1 << 20) + 1, 1)):   # single-tile kernels: half / quarter tiles, k_reduce_units
    model = getattr(cases, name)()
    ds = []
    for rep in range(4):
        pf = NativePf(model, n, 77); pf.set_option(3, fused)
        ll, ll_t, ess_t, _ = pf.run(t, y, has)
        ds.append(digest(np.array([ll]), ll_t, ess_t, pf.ancestors(), pf.particles()))
        pf.close()
    ok = len(set(ds)) == 1; bad += not ok
    print(f"{name} N={n} fused={fused}: {'identical' if ok else 'DIFFERENT'} {ds[0]} ll={ll:.6f}", flush=True)
for n, world in (((1 << 21) + 12, 4), (1 << 20, 8)):
    model = cases.c2_model()
    ds = []
    for rep in range(3):
        shards = [GpuShard(model, n, r, world, 77, 0) for r in range(world)]
        f = ShardedFilter(shards, LocalComm(world))
        ll, ess = f.ll_filter(t[:120], y[:120], has[:120])
        ds.append(digest(np.array([ll, ess]), np.concatenate([s.particles() for s in shards], axis=1)))
        assert f.last_single and f.last_attempts == 1, (f.last_single, f.last_attempts)
        for s in shards: s.close()
    ok = len(set(ds)) == 1; bad += not ok
    print(f"sharded N={n} world={world}: {'identical' if ok else 'DIFFERENT'} {ds[0]} ll={ll:.6f}", flush=True)
    pf = NativePf(model, n, 77); l1 = pf.run(t[:120], y[:120], has[:120])[0]; pf.close()
    print(f"   single-GPU ll of the same series: {l1:.6f} {'== sharded' if l1 == ll else '!= sharded'}", flush=True); bad += (l1 != ll)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)

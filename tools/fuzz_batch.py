"""Ad-hoc fuzz of the batched filters (cssm_pfb_*: B chains per launch) against handles of their own: random B, cloud sizes (all launch
geometries), per-chain parameters and seeds, series with missing and -- sometimes -- outlying observations, the batch run twice.
usage (GPU box): python tools/fuzz_batch.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf, NativePfBatch

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(ncases):
    B = int(rng.integers(2, 10))
    n = int(rng.choice([int(rng.integers(64, 5000)), int(rng.integers(5000, 200_000)), int(rng.integers(64 * 1024 + 1, 300 * 1024))]))
    whole = int(rng.integers(0, 3))
    T = int(rng.integers(4, 12))
    which = ["c2", "c1"][int(rng.integers(0, 2))]
    if which == "c2":
        um, mk = cases.c2_unparam(), cases.c2_params
    else:
        from composablestatespacemodels_amd import Model, Parameters, Sde, SdeParameter
        um, mk = Model.poisson(Sde.brownianMotion(1)), (lambda: Parameters.apply(None, SdeParameter.brownianParameter(0.0, 1.0, 0.01)))
    models = []
    for k in range(B):
        p = mk(); th = np.asarray(p.flattenParams())
        models.append(um.run(p.withFlat(th + 0.05 * rng.standard_normal(th.size))))
    seeds = [int(rng.integers(1, 1 << 40)) for _ in range(B)]
    t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
    y = y.copy()
    if rng.random() < 0.4:
        s = int(rng.integers(1, T)); y[s] = 55.0; has[s] = 1
    b = NativePfBatch(models[0], n, B)
    if whole:
        for k in range(B): b.chain(k).set_option(6, whole)
    ok = True
    for rep in range(2):
        ll, path, rc = b.filter(models, seeds, t, y, has)
        for k in range(B):
            g = NativePf(models[k], n, seeds[k]); g.set_option(6, whole)
            gl, _, _, gpath = g.run(t, y, has, want_path=True)
            ok &= bool(rc[k] == 0 and ll[k] == gl and np.array_equal(path[k], gpath) and np.array_equal(b.chain(k).particles(), g.particles()))
            g.close()
        seeds = seeds[1:] + seeds[:1]
    print(f"case {c}: {which} B={B} N={n} whole={whole} T={T}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    bad += 0 if ok else 1
    b.close()
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
sys.exit(1 if bad else 0)

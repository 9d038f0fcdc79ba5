"""Ad-hoc fuzz of the group-sum path (Scalars::grp) against the oracle: random cloud sizes with 64 .. 300 units of 1024 particles
(CSSM_OPT_WHOLE_TILES = 1 / 2 force one block per unit at these sizes), random series with missing observations and an outlier.
usage (GPU box): python tools/fuzz_group_sums.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(ncases):
    n = int(rng.integers(64 * 1024 + 1, 300 * 1024))
    whole = int(rng.integers(1, 3))
    name = ["c2_model", "c1_model", "c3_model"][int(rng.integers(0, 3))]
    model = getattr(cases, name)()
    T = int(rng.integers(4, 9))
    t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
    y = y.copy()
    if rng.random() < 0.5:
        s = int(rng.integers(1, T)); y[s] = 60.0; has[s] = 1
    seed = int(rng.integers(1, 1 << 40))
    o = oracle.OraclePf(model.descriptor(), n, seed)
    ol, oll, oess, _ = o.filter(t, y, has)
    g = NativePf(model, n, seed); g.set_option(6, whole)
    ok = True
    for rep in range(2):
        gl, gll, gess, _ = g.run(t, y, has)
        ok &= (gl == ol) and np.array_equal(gll, oll) and np.array_equal(gess, oess) and np.array_equal(g.ancestors(), o.ancestors()) \
            and np.array_equal(g.particles(), o.particles())
    g.close()
    print(f"case {c}: {name} N={n} whole={whole} T={T}: {'identical' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} mismatches")
sys.exit(1 if bad else 0)

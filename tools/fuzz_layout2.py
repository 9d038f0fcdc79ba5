"""Ad-hoc fuzz of the large-cloud geometry of round 5 (units of 4 tiles, up to 4096 units, group sums in layout 2: 64 groups x 64 units):
random cloud sizes between 4.2 and 20 million particles (whole and ragged), d = 1 / 3 models, series with missing observations, an outlier
(redone relative to the max: the lean path on group sums) and a continued part -- three handles must agree bit for bit: the default, the
group sums off (CSSM_OPT_GROUP_SUMS = 0), and the units of the rule before round 5 (CSSM_UNIT_MAX_TILES = 0).  The one-thread oracle
cannot follow at these sizes; slices of it can (tests/test_gpu_parity.py::test_full_size_configs_oracle_checked_on_slices).
usage (GPU box): python tools/fuzz_layout2.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for c in range(ncases):
    n = int(rng.integers(4_200_000, 20_000_000))
    if rng.random() < 0.3:
        n = (n >> 12) << 12                      # whole units now and then
    name = ["c2_model", "c1_model"][int(rng.integers(0, 2))]
    model = getattr(cases, name)()
    T = int(rng.integers(5, 9))
    t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
    y = y.copy()
    if rng.random() < 0.6:
        s = int(rng.integers(1, T - 1)); y[s] = 60.0; has[s] = 1
    seed = int(rng.integers(1, 1 << 40))
    cut = int(rng.integers(2, T))
    outs = []
    for grp, old_units in ((1, False), (0, False), (1, True)):
        if old_units:
            os.environ["CSSM_UNIT_MAX_TILES"] = "0"
        else:
            os.environ.pop("CSSM_UNIT_MAX_TILES", None)
        g = NativePf(model, n, seed); g.set_option(7, grp)
        l1 = g.run(t[:cut], y[:cut], has[:cut])
        l2 = g.run_more(t[cut:], y[cut:], has[cut:])
        outs.append((l1[0], l2[0], np.concatenate([l1[1], l2[1]]), np.concatenate([l1[2], l2[2]]), g.ancestors(), g.particles()))
        g.close()
    os.environ.pop("CSSM_UNIT_MAX_TILES", None)
    ok = True
    for other in outs[1:]:
        ok &= outs[0][0] == other[0] and outs[0][1] == other[1]
        for a, b in zip(outs[0][2:], other[2:]):
            ok &= bool(np.array_equal(a, b))
    print(f"case {c}: {name} N={n} T={T} cut at {cut}: {'identical' if ok else 'MISMATCH'} (ll {outs[0][1]!r})", flush=True)
    bad += (not ok)
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} mismatches")
sys.exit(1 if bad else 0)

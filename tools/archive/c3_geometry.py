"""(round 6) BASELINE configs[2] (d = 9, N = 2^22) under the launch geometries CSSM_OPT_WHOLE_TILES selects (0 = automatic: one tile per block + k_reduce_units;
1 = a unit per block, tile after tile; 2 = software-pipelined) and, on geometry 1, with the wave-range propagate + k_offspring_wave forced
(CSSM_OPT_WAVE_SUMS = 2): us per observation of a 300-observation series, best of three, and the kernels' event times.  usage: python tools/c3_geometry.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
n, T = 1 << 22, 300
t, y, has = cases.poisson_counts(T)
ref = None
for whole, ws in ((0, 1), (1, 0), (1, 2), (2, 0)):
    pf = NativePf(cases.c3_model(), n, cases.SEED); pf.set_option(6, whole); pf.set_option(10, ws)
    pf.run(t[:8], y[:8], has[:8])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ll, _, ess, _ = pf.run(t, y, has); best = min(best, time.perf_counter() - t0)
    pf.profile(True); pf.run(t, y, has); prof = pf.profile_read(); pf.profile(False)
    k = {a: round(v[0] / max(v[1], 1) * 1e3, 1) for a, v in prof.items() if v[1]}
    if ref is None: ref = (ll, ess.copy())
    same = ll == ref[0] and np.array_equal(ess, ref[1])
    print(f"geometry {whole} wave sums {ws}: {best / T * 1e6:7.1f} us per observation, kernels {k}, bits {'identical' if same else 'DIFFERENT'}", flush=True)
    pf.close()

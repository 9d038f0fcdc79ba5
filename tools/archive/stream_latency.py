"""Wall time of one streaming step (cssm_pf_step: record up, two kernels, ll / ess back, one synchronisation): usage stream_latency.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
t, y, has = cases.poisson_counts(400)
for n in (1000, 100000, 1 << 20):
    g = NativePf(cases.c2_model(), n, cases.SEED)
    g.init(0.0)
    for s in range(50): g.step(t[s], y[s], True)
    w = []
    for s in range(50, 350):
        t0 = time.perf_counter(); g.step(t[s], y[s], True); w.append((time.perf_counter() - t0) * 1e6)
    print(f"N={n}: streaming step wall {np.median(w):.1f} us (p10 {np.percentile(w, 10):.1f}, p90 {np.percentile(w, 90):.1f})")
    g.close()

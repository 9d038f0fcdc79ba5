"""Host time of one batch run beyond its device time (records built, uploaded, results read back): usage host_overhead.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
t, y, has = cases.poisson_counts(500)
for n in (1000, 100000):
    g = NativePf(cases.c2_model(), n, cases.SEED)
    g.run(t, y, has, want_path=False)
    w = []; l = []
    for _ in range(20):
        t0 = time.perf_counter(); g.run(t, y, has, want_path=False); w.append((time.perf_counter() - t0) * 1e3); l.append(g.last_loop_ms())
    print(f"N={n}: wall {np.median(w):.3f} ms, device loop {np.median(l):.3f} ms, difference {np.median(w) - np.median(l):.3f} ms per series of 500")
    g.close()

import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
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

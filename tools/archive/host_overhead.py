"""Host time of one batch run beyond its device time (records built, uploaded, results read back), with and without the
sampled path, and of one iteration of the native PMMH loop: usage host_overhead.py"""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
from composablestatespacemodels_amd.pmmh import pmmh_native
t, y, has = cases.poisson_counts(500)
for n in (1000, 100000):
    g = NativePf(cases.c2_model(), n, cases.SEED)
    for path in (False, True):
        g.run(t, y, has, want_path=path)
        w = []; l = []
        for _ in range(20):
            t0 = time.perf_counter(); g.run(t, y, has, want_path=path); w.append((time.perf_counter() - t0) * 1e3); l.append(g.last_loop_ms())
        print(f"N={n} path={path}: wall {np.median(w):.3f} ms, device loop {np.median(l):.3f} ms, difference {np.median(w) - np.median(l):.3f} ms per series of 500")
    g.close()
    um, p0 = cases.c2_unparam(), cases.c2_params()
    from composablestatespacemodels_amd import Data
    data = [Data(float(a), float(b) if c else None) for a, b, c in zip(t, y, has)]
    pmmh_native(um, p0, data, n, 0.05, 3, seed=11)
    t0 = time.perf_counter()
    r = pmmh_native(um, p0, data, n, 0.05, 30, seed=11)
    print(f"N={n}: native PMMH loop {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per iteration (incl. handle creation / 30)")

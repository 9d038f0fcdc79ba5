"""Host overhead of one short continued leg (bench.py --steps 20): wall time of run_more(K) against its device loop time."""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases, torch
from composablestatespacemodels_amd.filter import NativePf
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t, y, has = cases.poisson_counts(40 * K + 10)
g = NativePf(cases.c2_model(), 1 << 20, cases.SEED)
g.run(t[:5], y[:5], has[:5])
w = []; l = []
for r in range(30):
    lo = 5 + r * K
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
    torch.cuda.synchronize(); w.append((time.perf_counter() - t0) * 1e6); l.append(g.last_loop_ms() * 1e3)
print(f"K={K}: wall {np.median(w):.1f} us, device loop {np.median(l):.1f} us, host overhead {np.median(w) - np.median(l):.1f} us per leg = {(np.median(w) - np.median(l)) / K:.2f} us per step")
g.close()

import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["CSSM_CALL_TIMING"] = "1"; os.environ["CSSM_LOOP_EVENTS"] = "1"
import numpy as np, torch, cases
from composablestatespacemodels_amd.filter import NativePf
K = 20
model = cases.c2_model(); t, y, has = cases.poisson_counts(5 + 60 * K)
g = NativePf(model, 1 << 20, cases.SEED)
g.run(t[:5], y[:5], has[:5])
lo = 5
for r in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K]); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    lo += K
    if r >= 30: print(f"leg {r}: run_more {(t1 - t0) * 1e6:.1f} us, closing sync {(t2 - t1) * 1e6:.1f} us, total {(t2 - t0) * 1e6:.1f}", flush=True)

"""Wall time of short continued legs (bench.py --steps K's timed region), events off: python tools/leg_wall.py [K] [N]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases, torch
from composablestatespacemodels_amd.filter import NativePf
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
t, y, has = cases.poisson_counts(60 * K + 10)
g = NativePf(cases.c2_model(), n, cases.SEED)
g.run(t[:5], y[:5], has[:5])
w = []
for r in range(50):
    lo = 5 + r * K
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
    torch.cuda.synchronize(); w.append((time.perf_counter() - t0) * 1e6)
print(" ".join(f"{x:.0f}" for x in w)); w = np.array(w[10:])
print(f"K={K} N={n}: wall median {np.median(w):.1f} us, min {w.min():.1f} = {np.median(w) / K:.2f} us per step")
g.close()

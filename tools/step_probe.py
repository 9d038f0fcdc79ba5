"""Device time per observation of a batch run (cssm_pf_last_loop_ms / T) for a grid of sizes and option settings.
usage: step_probe.py [model=c2|c1|d<k>] [T=200] [sizes]      prints N, series, fused, one_launch -> us per observation"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sizes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1000, 100_000, 1 << 18, 1 << 20, 1 << 24]
model = cases.c2_model() if which == "c2" else (cases.c1_model() if which == "c1" else cases.dim_model(int(which[1:])))   # d<k>: latent dimension k
for n in sizes:
    TT = T if n <= (1 << 20) else max(20, T // 8)
    t, y, has = cases.poisson_counts(TT)
    row = []
    for series, fused, one in ((0, 1, 1), (0, 1, 0), (0, 0, 0), (1, 1, 0)):
        g = NativePf(model, n, cases.SEED)
        g.set_option(4, series); g.set_option(3, fused); g.set_option(5, one)
        g.run(t[:20], y[:20], has[:20])
        best = 1e9
        for _ in range(3):
            g.run(t, y, has)
            best = min(best, g.last_loop_ms() * 1e3 / TT)
        used = g.series_phases()[0]
        row.append(f"series={int(used)} fused={fused} one_launch={one}: {best:7.2f} us")
        g.close()
    print(f"{which} N={n:9d} T={TT}: " + " | ".join(row), flush=True)

"""Per-step time of the batch filter at small N with separate vs fused sums (run on the GPU box)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from composablestatespacemodels_amd.filter import NativePf
m = cases.c2_model(); t, y, has = cases.poisson_counts(500)
for n in (1000, 10000, 100000, 131072, 262144, 524288, 1 << 20):
    row = []
    for fused in (0, 1):
        pf = NativePf(m, n, 1); pf.set_option(3, fused)
        pf.run(t[:50], y[:50], has[:50])
        best = 1e9
        for _ in range(3):
            pf.run(t, y, has); best = min(best, pf.last_loop_ms() / 500 * 1e3)
        pf.profile(True); pf.run(t[:100], y[:100], has[:100]); p = pf.profile_read(); pf.close()
        row.append((best, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in p.items() if v[1]}))
    print(f"N={n}: separate {row[0][0]:.1f} us/step {row[0][1]} | fused {row[1][0]:.1f} us/step {row[1][1]}", flush=True)

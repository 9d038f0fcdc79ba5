"""Separate the fixed cost of one batch call from the per-step cost (run on the GPU box)."""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import cases
from composablestatespacemodels_amd.filter import NativePf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
model = cases.c2_model()
t, y, has = cases.poisson_counts(800)
pf = NativePf(model, n, 1)
pf.run(t[:50], y[:50], has[:50])
for T in (1, 100, 200, 400, 800):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); pf.run(t[:T], y[:T], has[:T]); best = min(best, time.perf_counter() - t0)
    print(f"N={n} T={T}: wall {best*1e3:.2f} ms, device loop {pf.last_loop_ms():.2f} ms, wall/step {best*1e6/T:.1f} us")

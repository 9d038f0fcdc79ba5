"""Per-kernel event times (us) of the per-observation kernels for a size list: kernel_probe.py [c2|c1] [sizes] [fused]"""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 20, 1 << 24]
fused = int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = cases.c2_model() if which == "c2" else cases.c1_model()
for n in sizes:
    T = 200 if n <= (1 << 20) else 24
    t, y, has = cases.poisson_counts(T)
    g = NativePf(model, n, cases.SEED); g.set_option(3, fused)
    g.run(t[:20], y[:20], has[:20])
    best = min((g.run(t, y, has), g.last_loop_ms())[1] for _ in range(3)) * 1e3 / T
    g.profile(True); g.run(t, y, has); prof = g.profile_read(); g.profile(False)
    print(f"{os.environ.get('CSSM_PF_LIB','default')[-12:]} {which} N={n} fused={fused}: step {best:.2f} us | " +
          " ".join(f"{k}={v[0] / v[1] * 1e3:.2f}" for k, v in prof.items() if v[1]), flush=True)
    g.close()

"""Per-kernel event times of a batch run: usage kernel_times.py model N [T=60]   (model: c2 | c1 | d<k>)"""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import cases
from composablestatespacemodels_amd.filter import NativePf
which, n = sys.argv[1], int(sys.argv[2]); T = int(sys.argv[3]) if len(sys.argv) > 3 else 60
model = cases.c2_model() if which == "c2" else (cases.c1_model() if which == "c1" else cases.dim_model(int(which[1:])))
t, y, has = cases.poisson_counts(T)
g = NativePf(model, n, cases.SEED)
g.run(t[:10], y[:10], has[:10]); g.run(t, y, has)
loop = g.last_loop_ms() * 1e3 / T
g.profile(True); g.run(t, y, has); prof = g.profile_read()
print(which, n, f"step {loop:.1f} us;", {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if v[1]}, "CSSM_SPLIT_X=" + os.environ.get("CSSM_SPLIT_X", "-"))
g.close()

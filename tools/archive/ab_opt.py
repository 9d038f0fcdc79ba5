"""A/B of one library option in one process (two handles, alternating runs): usage ab_opt.py model N T option valueA valueB
(model: c2 | c1 | c4 (LGCP, precision 2, event times) | d<k>).  Prints the best and median device time per observation and checks that both settings give the same bits."""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
which, n, T, opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
prec = 2 if which == "c4" else 0
model = cases.c4_model() if which == "c4" else (cases.c2_model() if which == "c2" else (cases.c1_model() if which == "c1" else cases.dim_model(int(which[1:]))))
t, y, has = cases.event_times(T, horizon=0.1 * T) if which == "c4" else cases.poisson_counts(T, missing=0.05)
hs = {}
for v in (va, vb):
    g = NativePf(model, n, cases.SEED, lgcp_precision=prec); g.set_option(opt, v); g.run(t[:10], y[:10], has[:10]); hs[v] = g
res = {v: [] for v in hs}; out = {}
for rep in range(5):
    for v, g in hs.items():
        r = g.run(t, y, has); res[v].append(g.last_loop_ms() * 1e3 / T); out[v] = (r[0], r[1], r[2], g.ancestors())
same = out[va][0] == out[vb][0] and all(np.array_equal(a, b) for a, b in zip(out[va][1:], out[vb][1:]))
prof = {}
for v, g in hs.items():
    g.profile(True); g.run(t, y, has); p = g.profile_read(); prof[v] = {k: round(x[0] / max(x[1], 1) * 1e3, 1) for k, x in p.items() if x[1]}
for v in hs:
    print(f"{which} N={n} option {opt} = {v}: step {min(res[v]):7.2f} us (median {np.median(res[v]):7.2f})  {prof[v]}")
print("   results identical:", same)
for g in hs.values(): g.close()

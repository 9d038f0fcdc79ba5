"""A/B in one process: the single-tile propagate kernel + k_reduce_units (default for large clouds) against whole units and the
software-pipelined kernel (CSSM_OPT_WHOLE_TILES = 2).  usage: ab_fine.py model N [T=40] [optA optB]  (model: c2 | c1 | d<k>; opt = CSSM_OPT_WHOLE_TILES value: 0 default, 1 single-tile,
2 whole units + software-pipelined kernel, one block per unit)"""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")   # (cssm_pf_last_loop_ms needs the event pair: CSSM_OPT_LOOP_EVENTS)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
which, n = sys.argv[1], int(sys.argv[2]); T = int(sys.argv[3]) if len(sys.argv) > 3 else (40 if n >= (1 << 20) else 200)
model = cases.c2_model() if which == "c2" else (cases.c1_model() if which == "c1" else cases.dim_model(int(which[1:])))
t, y, has = cases.poisson_counts(T)
hs = {}
opts = [int(v) for v in sys.argv[4:6]] if len(sys.argv) > 5 else [1 if n >= (1 << 20) else 0, 2]   # below 2^20 the default geometry IS the single-tile kernel
labels = {0: "default", 1: "fine", 2: "pipelined"}
for name, whole in ((labels[opts[0]], opts[0]), (labels[opts[1]], opts[1])):
    g = NativePf(model, n, cases.SEED); g.set_option(6, whole); g.run(t[:10], y[:10], has[:10]); hs[name] = g
res = {k: {"loop": [], "prop": [], "off": [], "red": []} for k in hs}
for rep in range(4):
    for name, g in hs.items():
        g.profile(False); g.run(t, y, has); res[name]["loop"].append(g.last_loop_ms() * 1e3 / T)
        g.profile(True); g.run(t, y, has); p = g.profile_read(); g.profile(False)
        # profile_read accumulates: take the differences of totals
        res[name]["prop"].append(p["k_propagate"]); res[name]["off"].append(p["k_offspring"]); res[name]["red"].append(p["k_reduce_units"])
def per(v):
    out = []
    prev = (0.0, 0)
    for ms, cnt in v:
        if cnt > prev[1]: out.append((ms - prev[0]) / (cnt - prev[1]) * 1e3)
        prev = (ms, cnt)
    return min(out) if out else 0.0
for name in hs:
    r = res[name]
    print(f"{which} N={n} {name:9s}: step {min(r['loop']):7.1f} us (median {np.median(r['loop']):7.1f}); k_propagate {per(r['prop']):6.1f}  k_reduce_units {per(r['red']):4.1f}  k_offspring {per(r['off']):6.1f}  (event-bracketed, +~2.6 us each)")
for g in hs.values(): g.close()

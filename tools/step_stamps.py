"""(experiment) phase durations of k_step's block 0, from a library built with -DCSSM_STEP_STAMPS (CSSM_PF_LIB=...):
usage: step_stamps.py N [T=100]   prints the median of phases A (prologue), B (parents' end slots), C (run fill), D (propagate), E (sums)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
n = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t, y, has = cases.poisson_counts(T)
g = NativePf(cases.c2_model(), n, cases.SEED)
g.set_option(5, 1)
g.run(t, y, has)
ll, ll_t, ess_t, _ = g.run(t, y, has, want_path=False)
v = np.asarray(ll_t[:-1])
ph = np.stack([np.floor(v / 10.0 ** (3 * k)) % 1000 for k in range(5)], axis=1) * 0.01
print(f"N={n}: per-observation {g.last_loop_ms() * 1e3 / T:.2f} us; block 0 phases A..E (us, median):", np.median(ph, axis=0), "sum", np.median(ph.sum(axis=1)))
g.close()

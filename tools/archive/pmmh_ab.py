"""PMMH (C5: N = 100 000, T = 500) ms per MCMC iteration, three runs of 40 iterations: for same-box A/B of two builds (CSSM_PF_LIB)."""
import os, sys, time, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from composablestatespacemodels_amd.pmmh import pmmh_native
from composablestatespacemodels_amd import Data
t, y, has = cases.poisson_counts(500)
data = [Data(float(a), float(b)) for a, b in zip(t, y)]
for rep in range(3):
    t0 = time.perf_counter(); ll, th, acc, last = pmmh_native(cases.c2_unparam(), cases.c2_params(), data, 100000, 0.05, 40, seed=7)
    dt = time.perf_counter() - t0
    print(os.environ.get("CSSM_PF_LIB", "default")[-8:], "rep", rep, "ms/iter %.3f" % (dt / 40 * 1e3), "ll_last", float(ll[-1]), flush=True)

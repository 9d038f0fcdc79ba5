"""(experiment) phase durations of k_offspring_self's block 0, from a library built with -DCSSM_OFF_STAMPS (CSSM_PF_LIB=...):
usage: off_stamps.py N   prints the median of: first round trip + max decode, unit sums, tile (exp, fix, scan, end slots), run fill + stores."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
n = int(sys.argv[1]); T = 100
t, y, has = cases.poisson_counts(T)
g = NativePf(cases.c2_model(), n, cases.SEED)
g.run(t, y, has)
T2 = 150
ll_t = np.zeros(T2); ess_t = np.zeros(T2, dtype=np.int32)
tt = np.arange(T2, dtype=np.float64); yy = np.resize(y, T2); hh = np.ones(T2, dtype=np.uint8)
ll = ctypes.c_double()
lib = g.lib
rc = lib.cssm_pf_ll_filter(g._h, tt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), yy.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                           hh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_size_t(T2), ctypes.byref(ll),
                           ll_t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ess_t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
v = ll_t[:149]
ph = np.stack([np.floor(v / 10.0 ** (3 * k)) % 1000 for k in range(4)], axis=1) * 0.01
print(f"N={n}: rc={rc} block 0 phases (us, median):", np.median(ph, axis=0), "sum", np.median(ph.sum(axis=1)))
g.close()

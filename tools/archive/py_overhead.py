"""Where the Python side of a timed leg goes: idle torch.cuda.synchronize(), NativePf.run_more against the bare ctypes call."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import ctypes as C
import numpy as np, cases, torch
from composablestatespacemodels_amd.filter import NativePf
K = 20
t, y, has = cases.poisson_counts(200 * K + 10)
g = NativePf(cases.c2_model(), 1 << 20, cases.SEED)
g.run(t[:5], y[:5], has[:5])
def med(f, n=200):
    w = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); w.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(w[n // 5:]))
torch.cuda.synchronize()
print(f"idle torch.cuda.synchronize(): {med(torch.cuda.synchronize):.1f} us")
pos = [5]
def leg():
    lo = pos[0]; pos[0] += K
    g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
print(f"run_more (wrapper, K={K}): {med(leg, 60):.1f} us")
def leg_sync():
    leg(); torch.cuda.synchronize()
print(f"run_more + torch.cuda.synchronize(): {med(leg_sync, 60):.1f} us")
ll = C.c_double(); ll_t = np.empty(K); ess_t = np.empty(K, dtype=np.int32)
def raw():
    lo = pos[0]; pos[0] += K
    g._ll_filter_more_raw(g._h, t[lo:].ctypes.data, y[lo:].ctypes.data, has[lo:].ctypes.data, K, C.byref(ll), ll_t.ctypes.data, ess_t.ctypes.data)
print(f"bare ctypes call: {med(raw, 60):.1f} us")
g.close()

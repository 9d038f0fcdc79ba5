"""Where does a step of the series kernel go?  Every block stamps its phases; print, per stamp, the spread over blocks."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model = cases.c2_model()
t, y, has = cases.poisson_counts(T)
g = NativePf(model, n, cases.SEED)
g.run(t, y, has)
g.lib.cssm_pf_profile(g._h, 2)
g.run(t, y, has)
nb, ns = C.c_uint32(), C.c_uint32()
g.lib.cssm_pf_series_stamps(g._h, None, 0, C.byref(nb), C.byref(ns))
buf = np.zeros(nb.value * ns.value * 5, dtype=np.uint64)
rc = g.lib.cssm_pf_series_stamps(g._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, None, None)
assert rc == 0
st = buf.reshape(nb.value, ns.value, 5).astype(np.float64) * 0.01   # us
t0 = st[:, :, 0].min(axis=0)                                         # earliest start of each step
names = ["start", "end P", "end exchange", "end O", "end barrier"]
print(f"N={n} blocks={nb.value} steps={ns.value}; times in us relative to the earliest block's start of the step, median over steps 10..")
for q in range(5):
    rel = st[:, 10:, q] - t0[None, 10:]
    print(f"  {names[q]:13s} min over blocks {np.median(rel.min(axis=0)):7.2f}  median {np.median(np.median(rel, axis=0)):7.2f}  max {np.median(rel.max(axis=0)):7.2f}")
step = np.diff(t0[10:])
print("  step (start to start): median %.2f us" % np.median(step))
durP = st[:, 10:, 1] - st[:, 10:, 0]
print("  phase P per block: min %.2f median %.2f max %.2f" % (np.median(durP.min(axis=0)), np.median(durP), np.median(durP.max(axis=0))))
durO = st[:, 10:, 3] - st[:, 10:, 2]
print("  phase O per block: min %.2f median %.2f max %.2f" % (np.median(durO.min(axis=0)), np.median(durO), np.median(durO.max(axis=0))))
slow = np.argsort(-np.median(durP, axis=1))[:8]
print("  slowest blocks in P:", slow, np.median(durP, axis=1)[slow].round(1))

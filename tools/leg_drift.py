"""(round 6) Is the leg-to-leg drift of a fresh filter's 20-observation legs DATA (the cloud's degeneracy over the first observations) or TIME
(clocks)?  Two fresh handles run the same 40 legs one after the other; a third runs them after a 2-second pause; device time per leg from the
GPU's clock stamps.  usage (GPU box): python tools/leg_drift.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases, torch
from composablestatespacemodels_amd.filter import NativePf
K, L = 20, 40
t, y, has = cases.poisson_counts(5 + L * K + 10)
def legs(pause=0.0):
    g = NativePf(cases.c2_model(), 1 << 20, cases.SEED)
    torch.cuda.synchronize(); time.sleep(pause)
    g.run(t[:5], y[:5], has[:5])
    dev, ess = [], []
    for r in range(L):
        lo = 5 + r * K
        torch.cuda.synchronize()
        _, _, e = g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
        torch.cuda.synchronize()
        dev.append(g.last_device_us()); ess.append(float(np.mean(e)))
    g.close()
    return np.array(dev), np.array(ess)
a, ea = legs(); b, _ = legs(); c, _ = legs(2.0)
f = lambda v: " ".join("%5.0f" % x for x in v)
print("device us per leg, handle 1:", f(a)); print("device us per leg, handle 2:", f(b)); print("handle 3 (after 2 s idle):  ", f(c))
print("mean ESS / N per leg:       ", " ".join("%5.2f" % (x / (1 << 20)) for x in ea))
print("correlation of handle 1 and 2 per leg: %.3f; of device time with ESS: %.3f" % (np.corrcoef(a, b)[0, 1], np.corrcoef(a, ea)[0, 1]))

import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.model import TimedObservation
from composablestatespacemodels_amd.pmmh import pmmh_native_speculative
n, T = 100_000, 500
t, y, has = cases.poisson_counts(T)
data = [TimedObservation(float(a), float(v)) for a, v in zip(t, y)]
um, init = cases.c2_unparam(), cases.c2_params()
ll, th, acc, last = pmmh_native_speculative(um, init, data, n, 0.05 ** 2, 60, seed=2)
print("accepted after 60 iterations:", acc[-1], "ll tail", ll[-3:])

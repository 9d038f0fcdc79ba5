"""C5 (BASELINE configs[4]: PMMH, N = 100 000 particles, T = 500, the seasonal-Poisson model) -- MCMC iterations per second of ONE host
thread: a single chain (cssm_pmmh_run) against B chains in lockstep whose filters run as one batch (cssm_pmmh_run_batched).
usage: pmmh_batched.py [iters=20] [B list, comma separated = 1,2,4,8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.model import TimedObservation
from composablestatespacemodels_amd.pmmh import pmmh_native, pmmh_native_batched
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Bs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
n, T = 100_000, 500
t, y, has = cases.poisson_counts(T)
data = [TimedObservation(float(a), float(v)) for a, v in zip(t, y)]
um, init = cases.c2_unparam(), cases.c2_params()
pmmh_native(um, init, data, n, 0.05 ** 2, 3, seed=1)          # (code objects, clocks)
t0 = time.perf_counter(); pmmh_native(um, init, data, n, 0.05 ** 2, iters, seed=2); w1 = time.perf_counter() - t0
print(f"1 chain  (cssm_pmmh_run):          {w1 / iters * 1e3:7.2f} ms per iteration, {iters / w1:7.1f} iterations/s, {n * T * iters / w1 / 1e9:6.2f} G particle-steps/s")
from composablestatespacemodels_amd.pmmh import pmmh_native_speculative
pmmh_native_speculative(um, init, data, n, 0.05 ** 2, 2, seed=2)
t0 = time.perf_counter(); sp = pmmh_native_speculative(um, init, data, n, 0.05 ** 2, iters, seed=2); ws = time.perf_counter() - t0
ref = pmmh_native(um, init, data, n, 0.05 ** 2, iters, seed=2)
same = all(np.array_equal(a, b) for a, b in zip(sp, ref))
print(f"1 chain  (cssm_pmmh_run_speculative: two iterations per batch of three filters): {ws / iters * 1e3:7.2f} ms per iteration, {iters / ws:7.1f} iterations/s "
      f"({w1 / ws:4.2f}x), the chain identical to cssm_pmmh_run's: {same}")
for B in Bs:
    inits = [init] * B
    seeds = [100 + k for k in range(B)]
    pmmh_native_batched(um, inits, data, n, 0.05 ** 2, 2, seeds)
    t0 = time.perf_counter(); pmmh_native_batched(um, inits, data, n, 0.05 ** 2, iters, seeds); w = time.perf_counter() - t0
    print(f"{B} chains (cssm_pmmh_run_batched): {w / iters * 1e3:7.2f} ms per lockstep iteration, {B * iters / w:7.1f} iterations/s in total "
          f"({B * iters / w / (iters / w1):4.2f}x one chain), {B * n * T * iters / w / 1e9:6.2f} G particle-steps/s")

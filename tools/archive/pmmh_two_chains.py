"""C5 as the reference runs it: TWO chains side by side (examples/DetermineParameters.scala:68-69, mapAsync(2)).
Each chain is a handle of its own on a stream of its own, driven from its own thread; at N = 100 000 a filter step is
launch-latency bound, so the chains overlap on the GPU.  Run on the GPU box."""
import os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
from composablestatespacemodels_amd import Data
from composablestatespacemodels_amd.pmmh import pmmh_native

t, y, has = cases.poisson_counts(500)
data = [Data(float(a), float(b) if h else None) for a, b, h in zip(t, y, has)]
init = cases.c2_model().parameters()
ITERS, N = 30, 100000

def chain(seed, out):
    out.append(pmmh_native(cases.c2_unparam(), init, data, N, 0.05, ITERS, seed=seed))

chain(1, [])                                   # warm-up (library load, allocations)
for nchains in (1, 2, 4):
    outs = [[] for _ in range(nchains)]
    th = [threading.Thread(target=chain, args=(100 + c, outs[c])) for c in range(nchains)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    w = time.perf_counter() - t0
    assert all(len(o) == 1 for o in outs)
    print(f"{nchains} chain(s) x {ITERS} iterations, N={N}, T=500: {w / ITERS * 1e3:.2f} ms per iteration of every chain, "
          f"{nchains * ITERS / w:.1f} iterations/s in total, {nchains * N * 500 * ITERS / w / 1e9:.2f} G particle-steps/s", flush=True)

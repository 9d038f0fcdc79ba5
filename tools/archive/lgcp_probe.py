"""Device time per event of the single-GPU LGCP filter (configs[3]) and per observation of C1 / C2: python tools/lgcp_probe.py"""
import os; os.environ.setdefault("CSSM_LOOP_EVENTS", "1")
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
tag = os.environ.get("CSSM_PF_LIB", "default")[-8:]
for name, n, T, prec, data in (("c4", 1 << 21, 60, 2, cases.event_times), ("c4", 1 << 24, 24, 2, cases.event_times),
                               ("c1", 1 << 24, 24, 0, cases.poisson_counts), ("c2", 1 << 20, 200, 0, cases.poisson_counts)):
    model = getattr(cases, name + "_model")()
    t, y, has = data(T)
    g = NativePf(model, n, cases.SEED, lgcp_precision=prec)
    g.run(t[:8], y[:8], has[:8])
    best = min((g.run(t, y, has), g.last_loop_ms())[1] for _ in range(3)) * 1e3 / T
    print(f"{tag} {name} N={n}: {best:.2f} us per observation", flush=True)
    g.close()

#!/usr/bin/env python3
"""Run the five BASELINE.json configurations on one MI355X and print a markdown table + JSON.

C1 is the reference's CPU-runnable case (timed on the oracle AND on the GPU); C4 runs the 16M-particle
LGCP filter on ONE GPU here (the 8-GPU sharding is the driver's scaling run); C5 runs a short PMMH chain
and reports seconds per iteration (10 000 iterations = that x 10 000)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd import Data
from composablestatespacemodels_amd.filter import NativePf
from composablestatespacemodels_amd.pmmh import pmmh_native
from oracle import oracle

out = []

def run(name, model, n, t, y, has, prec=0, reps=2):
    pf = NativePf(model, n, cases.SEED, lgcp_precision=prec)
    pf.run(t[:8], y[:8], has[:8])
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); ll, _, ess, _ = pf.run(t, y, has); best = min(best, time.perf_counter() - t0)
    pf.profile(True); pf.run(t, y, has); prof = pf.profile_read(); pf.profile(False)
    k = {a: round(v[0] / max(v[1], 1) * 1e3, 1) for a, v in prof.items() if v[1]}
    d = pf.d
    row = {"config": name, "N": n, "T": len(t), "d": d, "wall_s": best, "particle_steps_per_s": n * len(t) / best,
           "us_per_step": best / len(t) * 1e6, "ll": ll, "ess_mean": float(np.mean(ess)), "kernels_us": k,
           "k_propagate_alg_GBs": (16 * d + 8) * n / (k["k_propagate"] * 1e-6) / 1e9}
    pf.close(); out.append(row); print(json.dumps(row), flush=True)
    return row

# C1
m = cases.c1_model(); t, y, has = cases.poisson_counts(100)
r = run("C1 poisson+brownian N=1k T=100", m, 1000, t, y, has)
o = oracle.OraclePf(m.descriptor(), 1000, cases.SEED)
t0 = time.perf_counter(); oll = o.filter(t, y, has)[0]; r["cpu_oracle_s"] = time.perf_counter() - t0; r["cpu_ll_equal"] = (oll == r["ll"])
# C2
m = cases.c2_model(); t, y, has = cases.poisson_counts(500)
run("C2 seasonal-poisson OU d=3 N=2^20 T=500", m, 1 << 20, t, y, has)
# C3
m = cases.c3_model(); t, y, has = cases.poisson_counts(1000)
run("C3 poisson+seasonal(4 harmonics) d=9 N=2^22 T=1000", m, 1 << 22, t, y, has, reps=1)
# C4 (single GPU share and the full 16M on one GPU)
m = cases.c4_model(); t, y, has = cases.event_times(100)
run("C4 LGCP precision 2, N=2^21 (one of 8 shards), 100 events on [0,10]", m, 1 << 21, t, y, has, prec=2, reps=1)
run("C4 LGCP precision 2, N=2^24 on ONE GPU, 100 events", m, 1 << 24, t, y, has, prec=2, reps=1)
# C5
t, y, has = cases.poisson_counts(500)
data = [Data(float(a), float(b)) for a, b in zip(t, y)]
for n in (100000, 131072):
    pmmh_native(cases.c2_unparam(), cases.c2_params(), data, n, 0.05, 2, seed=7)   # (untimed: the first call of a process loads the code objects)
    t0 = time.perf_counter(); ll, th, acc, last = pmmh_native(cases.c2_unparam(), cases.c2_params(), data, n, 0.05, 20, seed=7)
    dt = time.perf_counter() - t0
    # ... and the same chain at two iterations per batch of three filters (cssm_pmmh_run_speculative: identical output)
    from composablestatespacemodels_amd.pmmh import pmmh_native_speculative
    pmmh_native_speculative(cases.c2_unparam(), cases.c2_params(), data, n, 0.05, 2, seed=7)
    t0 = time.perf_counter(); sp = pmmh_native_speculative(cases.c2_unparam(), cases.c2_params(), data, n, 0.05, 20, seed=7)
    ds = time.perf_counter() - t0
    row = {"config": f"C5 PMMH seasonal model N={n} T=500", "iters_run": 20, "s_per_iter": dt / 20, "projected_10k_iters_s": dt / 20 * 10000,
           "particle_steps_per_s": n * 500 * 20 / dt, "accepted": int(acc[-1]), "ll_last": float(ll[-1]),
           "speculative_s_per_iter": ds / 20, "speculative_projected_10k_iters_s": ds / 20 * 10000,
           "speculative_identical": bool(all(np.array_equal(a, b) for a, b in zip(sp, (ll, th, acc, last))))}
    out.append(row); print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "configs.json"), "w"), indent=1)

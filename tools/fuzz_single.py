"""Ad-hoc fuzz of the single-GPU drivers against the oracle: random model, cloud size (1 .. 300 000: every launch geometry), resampler,
series with missing and outlying observations, cut at random places into cssm_pf_ll_filter / _ll_filter_more calls and streaming steps.
usage (GPU box): python tools/fuzz_single.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NAMES = ["c1_model", "c2_model", "c3_model", "linear_model", "negbin_model", "euler_model", "c4_model", "lgcp_seasonal_model"]
bad = 0
for c in range(ncases):
    name = NAMES[int(rng.integers(0, len(NAMES)))]
    lg = name in ("c4_model", "lgcp_seasonal_model")
    n = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, 70000)), int(rng.integers(70000, 300000))]))
    T = int(rng.integers(6, 20))
    kind = 0 if lg else int(rng.choice([0, 0, 1, 2]))
    whole = int(rng.integers(0, 4))
    model = getattr(cases, name)()
    if lg:
        t, y, has = cases.event_times(T, horizon=float(rng.uniform(5.0, 12.0)))
    else:
        t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
        y = y.copy()
        if name == "linear_model": y = y - 1.0
        elif rng.random() < 0.4:
            s = int(rng.integers(1, T)); y[s] = float(rng.choice([35.0, 60.0, 90.0])); has[s] = 1
    prec = 2 if lg else 0
    flags = {0: 0, 1: oracle.RESAMPLE_STRATIFIED, 2: oracle.RESAMPLE_MULTINOMIAL}[kind]
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED, flags)
    ol, oll, oess, _ = o.filter(t, y, has)
    g = NativePf(model, n, cases.SEED, lgcp_precision=prec); g.set_option(6, whole); g.set_option(2, kind)
    a = int(rng.integers(1, T - 2)); b = int(rng.integers(a + 1, T))
    ll_t, ess_t = [], []
    r = g.run(t[:a], y[:a], has[:a]); ll_t.append(r[1]); ess_t.append(r[2])
    if rng.random() < 0.5:
        r = g.run_more(t[a:b], y[a:b], has[a:b]); ll_t.append(r[1]); ess_t.append(r[2])
    else:
        for s in range(a, b):
            l1, e1 = g.step(t[s], y[s], bool(has[s])); ll_t.append([l1]); ess_t.append([e1])
    r = g.run_more(t[b:], y[b:], has[b:]); ll_t.append(r[1]); ess_t.append(r[2])
    gll, gess = np.concatenate(ll_t), np.concatenate(ess_t)
    ok = r[0] == ol and np.array_equal(gll, oll) and np.array_equal(gess, oess) and np.array_equal(g.particles(), o.particles()) and np.array_equal(g.ancestors(), o.ancestors())
    print(f"case {c}: {name} N={n} T={T} cuts {a},{b} resampler={kind} whole={whole}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    bad += 0 if ok else 1
    g.close()
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
sys.exit(1 if bad else 0)

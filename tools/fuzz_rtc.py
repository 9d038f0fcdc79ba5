"""Ad-hoc fuzz of the run-time structure specialisation (csrc/cssm_rtc.cpp): randomly composed models -- a count / Gaussian observation
leaf on a random SDE, zero to two seasonal leaves on random SDEs of random dimensions (d <= 9) -- each through the kernel compiled for
its structure AND the structure-as-data kernel, against the oracle, in two launch geometries.
usage (GPU box): python tools/fuzz_rtc.py [cases] [seed]"""
import os, sys
import ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd import Model, Parameters, Sde, SdeParameter, _abi
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

def info():
    out = (C.c_uint64 * 4)(); _abi.check(_abi.load_library().cssm_rtc_info(out)); return list(out)

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

def sde(dim):
    k = int(rng.integers(0, 3))
    m0 = list(rng.normal(0.0, 0.3, dim)); c0 = 1.0
    if k == 0: return Sde.brownianMotion(dim), SdeParameter.brownianParameter(m0 if dim > 1 else m0[0], c0, float(rng.uniform(0.005, 0.05))), "bm"
    if k == 1: return Sde.genBrownianMotion(dim), SdeParameter.genBrownianParameter(m0 if dim > 1 else m0[0], c0, float(rng.normal(0, 0.02)), float(rng.uniform(0.005, 0.05))), "gbm"
    return Sde.ouProcess(dim), SdeParameter.ouParameter(m0 if dim > 1 else m0[0], c0, float(rng.uniform(0.1, 0.4)), list(rng.normal(0, 0.3, dim)) if dim > 1 else float(rng.normal(0, 0.3)), float(rng.uniform(0.1, 0.4))), "ou"

bad = 0
for c in range(ncases):
    s0, p0, n0 = sde(1)
    obs = int(rng.integers(0, 3))
    if obs == 0: m, p, name = Model.poisson(s0), Parameters.apply(None, p0), "poisson(" + n0 + ")"
    elif obs == 1: m, p, name = Model.negativeBinomial(s0), Parameters.apply(float(np.log(rng.uniform(1.5, 5.0))), p0), "negbin(" + n0 + ")"
    else: m, p, name = Model.linear(s0), Parameters.apply(float(np.log(rng.uniform(0.3, 1.5))), p0), "linear(" + n0 + ")"
    for _ in range(int(rng.integers(0, 3))):
        h = int(rng.integers(1, 4)); sd_, pd_, nm = sde(2 * h)
        m = m | Model.seasonal(int(rng.choice([7, 12, 24])), h, sd_); p = p | Parameters.apply(None, pd_); name += f" |+| seasonal(h={h},{nm})"
        if m.run(p).dimension > 9: break
    model = m.run(p)
    if model.dimension > 12: continue
    n = int(rng.choice([int(rng.integers(500, 6000)), int(rng.integers(64 * 1024 + 1, 100 * 1024))]))
    whole = int(rng.integers(0, 3))
    T = int(rng.integers(4, 9))
    t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.15)
    if obs == 2: y = y - 1.5
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED); ol, oll, oess, _ = o.filter(t, y, has)
    i0 = info()
    g = NativePf(model, n, cases.SEED); g.set_option(6, whole)
    gl, gll, gess, _ = g.run(t, y, has)
    i1 = info()
    ok = gl == ol and np.array_equal(gess, oess) and np.array_equal(g.ancestors(), o.ancestors()) and np.array_equal(g.particles(), o.particles())
    g.set_option(8, 0)
    ok &= g.run(t, y, has)[0] == ol
    print(f"case {c}: {name} d={model.dimension} N={n} whole={whole}: rtc launches {i1[2] - i0[2]}/{T} failures {i1[3]}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    bad += 0 if ok else 1
    g.close()
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
sys.exit(1 if bad else 0)

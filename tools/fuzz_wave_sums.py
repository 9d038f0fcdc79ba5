"""(round 6) Fuzz of k_offspring_wave + the wave-range propagate: random cloud sizes from 64 K to 20 M particles (whole and ragged), d = 1 / 3 / 9
models, series with missing observations, sometimes an outlying one (held and redone by the 128-bit kernels in place) or a level far above the
max, a continued part -- CSSM_OPT_WAVE_SUMS = 2 (wherever the geometry allows) against 0 (k_offspring_self) bit for bit: ll after every
observation, ESS, ancestors, the cloud; now and then with every chunk forced through the exact path; clouds up to 300 K also against the oracle.
usage (GPU box): python tools/fuzz_wave_sums.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
bad = 0
for c in range(ncases):
    big = rng.random() < 0.4
    n = int(rng.integers(1 << 20, 20_000_000)) if big else int(rng.integers(64 * 1024, 1 << 20))
    if rng.random() < 0.3:
        n = (n >> 12) << 12
    name = ["c2_model", "c1_model", "c3_model"][int(rng.integers(0, 3))]
    if name == "c3_model" and n > (1 << 21):
        name = "c1_model"
    model = getattr(cases, name)()
    T = int(rng.integers(4, 9))
    t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
    y = y.copy()
    kind = rng.random()
    if kind < 0.3:
        s = int(rng.integers(1, T)); y[s] = 70.0; has[s] = 1
    elif kind < 0.5:
        s = int(rng.integers(1, T)); y[s] = float(rng.integers(50, 66)); has[s] = 1
    seed = int(rng.integers(1, 1 << 40))
    cut = int(rng.integers(1, T))
    exact = int(rng.integers(0, 3)) if rng.random() < 0.25 else 0
    whole = 0 if n >= (1 << 20) else 1

    def run(ws, ex):
        g = NativePf(model, n, seed); g.set_option(10, ws); g.set_option(1, ex)
        if whole:
            g.set_option(6, whole)
        _, a, b, _ = g.run(t[:cut], y[:cut], has[:cut])
        ll, c2, d2 = g.run_more(t[cut:], y[cut:], has[cut:])
        out = (ll, np.concatenate([a, c2]), np.concatenate([b, d2]), g.ancestors(), g.particles())
        g.close()
        return out
    new, old = run(2, exact), run(0, 0)
    ok = new[0] == old[0] and all(np.array_equal(p, q) for p, q in zip(new[1:], old[1:]))
    if ok and n <= 300_000:
        o = oracle.OraclePf(model.descriptor(), n, seed)
        ol, oll, oess, _ = o.filter(t, y, has)
        ok = new[0] == ol and np.array_equal(new[1], oll) and np.array_equal(new[2], oess) and np.array_equal(new[3], o.ancestors())
    print(f"case {c}: {name} N={n} T={T} cut {cut} exact {exact}: {'identical' if ok else 'DIFFERENT'} (ll {new[0]!r})", flush=True)
    bad += 0 if ok else 1
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""A SECOND, independent statement of the hot path in numpy, and the fixture it emits ("a numpy mirror of the restatement
... to cross-check the C restatement").  Run in the BUILD container:

    python tests/golden/make_literal.py            # writes tests/golden/literal_runs.json

What is independent here.  Everything below the line "the literal path" is written from the reference's Scala lines (cited per function;
paths relative to src/main/scala/com/github/jonnylaw/model/) with numpy / libm arithmetic ONLY: it includes neither
include/cssm_numerics.h nor oracle/cssm_oracle.c, shares no elementary function, no fixed-point sum, no tie rule and no clamp with them.
Sums are the reference's sequential fp64 sums (`foldLeft`, `scanLeft`, `Seq.sum`), the ancestor search is `numpy.searchsorted(C, k,
side='left')` for `TreeMap.from(k).head` followed by the TreeMap's duplicate-key rule (the LAST particle inserted under a key wins,
Resampling.scala:57), the Poisson density is breeze's `-lambda + k log(lambda) - lgamma(k + 1)` with `lambda = exp(gamma)` as written, the
other seven observation densities are scipy.stats' norm / t / beta log-pdfs and `gammaln` forms of the Scala lines (data_likelihood).

What is shared, and why.  The reference draws its variates from unseeded global generators (breeze `rand.gaussian`, `scala.util.Random`):
no run of it can be reproduced even by itself.  A witness of the ARITHMETIC therefore needs the variates handed in: the standard normals
of every particle and the one uniform of every resampling are DUMPED from the oracle (oracle_pf_dump_normals / oracle_c_u, in its
LITERAL_SUMS | LIBM | TIE_LAST mode: Philox bits through libm's log / sqrt / cos / sin) and read here as data -- `z()` and `u` of
SURVEY.md Appendix A.  Nothing else crosses.

The fixture: for BASELINE's configs[0..3] at N <= 4096, for one run of every other observation model and transition at N = 2048 and for
the filter constructed with stratifiedResampling / multinomialResampling (per-slot uniforms dumped: oracle_c_strat_u_v, oracle_c_multi_u_v), the per-observation log-likelihood and ESS, the ancestors of the first and of the
last weighted observation and the final cloud's first component -- twice: `tie_last` is the reference's behaviour (TreeMap duplicate keys),
`tie_first` the same statement with the canonical lower bound (the build's deviation D3: the two differ only where a weight underflowed
to exactly 0, and are different realisations of the same estimator from the first such swap on).  tests/test_literal_mirror.py holds the CPU oracle (literal mode) and
-- on the GPU -- the HIP path (contract mode, through the C ABI) against it, with the tolerances stated there.

PMMH: `literal_mh` states MetropolisHastings.mhStep (PMMH.scala:68-81,121; Parameters.scala:65-67) over the numpy filter
above; the proposal's normals, the accept uniform and the key of every filter run are dumped (oracle_c_mh_normals, oracle_c_mh_u,
oracle_c_derive_key).  The fixture's chain (12 iterations) has accepted and rejected proposals.
"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np
from scipy.special import gammaln

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

SDE_BROWNIAN, SDE_GEN_BROWNIAN, SDE_OU, SDE_EULER_AFFINE = 0, 1, 2, 3


# ======================================================================================================= the literal path
def logistic(x):
    """SdeParameter.logistic, SdeParameters.scala:214-216."""
    return 1.0 / (1.0 + np.exp(-x))


def repeat_to(v, dim):
    """Sde.buildParamRepeat, Sde.scala:177-179: a parameter vector cyclically repeated to the leaf's dimension."""
    v = np.asarray(v, dtype=np.float64)
    return v[np.arange(dim) % len(v)]


class Leaf:
    """One leaf of the composed model with its CONSTRAINED parameters, as the SDE case classes hold them:
    BrownianMotion Sde.scala:98-102, GenBrownianMotion :69-73, OuProcess :129-137 (logistic applied to the stored phi, which the
    user-facing constructor ouParameter has already passed through logistic: SdeParameters.scala:204)."""

    def __init__(self, kind, dim, stored, seasonal=None):
        self.kind, self.dim, self.seasonal = kind, dim, seasonal   # seasonal = (period, harmonics) or None
        self.m0 = repeat_to(stored["m0"], dim)
        self.c0 = np.exp(repeat_to(stored["c0"], dim))
        if kind == SDE_BROWNIAN:
            self.sigma = np.exp(repeat_to(stored["sigma"], dim))
        elif kind == SDE_GEN_BROWNIAN:
            self.mu = repeat_to(stored["mu"], dim)
            self.sigma = np.exp(repeat_to(stored["sigma"], dim))
        elif kind == SDE_OU:
            self.phi = logistic(repeat_to(stored["phi"], dim))
            self.mu = repeat_to(stored["mu"], dim)
            self.sigma = np.exp(repeat_to(stored["sigma"], dim))
        else:   # a user-defined Sde on the trait's default Euler-Maruyama step: drift a + b x, diagonal diffusion g (raw values)
            self.a, self.b, self.g = repeat_to(stored["a"], dim), repeat_to(stored["b"], dim), repeat_to(stored["g"], dim)

    def initial(self, z):
        """initialState: diag(sqrt(c0)) * z + m0 -- Sde.scala:104-108 (Brownian), :75-80, :152-156.  z: [N, dim]."""
        return np.sqrt(self.c0) * z + self.m0

    def step(self, x, dt, z):
        """stepFunction(dt)(x): the exact Gaussian transitions of the three built-ins (Sde.scala:114-123, :86-95, :139-150) and the
        trait default stepEulerMaruyama (:36-43 with dW = sqrt(dt) z, :30-33).  x, z: [N, dim]."""
        if self.kind == SDE_BROWNIAN:
            return np.sqrt(self.sigma * dt) * z + x                                    # varianceMatrix * rand + mean, mean = x
        if self.kind == SDE_GEN_BROWNIAN:
            mean = x + self.mu * dt
            return np.sqrt(self.sigma * dt) * z + mean
        if self.kind == SDE_OU:
            variance = (self.sigma * self.sigma / (self.phi * 2.0)) * (1.0 - np.exp(self.phi * -2.0 * dt))   # :139-140
            mean = self.mu + (x - self.mu) * np.exp(-self.phi * dt)                                          # :144
            return np.sqrt(variance) * z + mean                                                              # :146
        dW = math.sqrt(dt) * z
        return (x + (self.a + self.b * x) * dt) + self.g * dW

    def f(self, x, t):
        """The leaf's f(s, t): x(0) (Model.scala:271 and its siblings) or buildF(harmonics, t) dot x (:217-225)."""
        if self.seasonal is None:
            return x[:, 0].copy()
        period, harmonics = self.seasonal
        frequency = 2 * math.pi / period
        acc = None
        for a in range(1, harmonics + 1):      # flatMap(a => Array(cos(frequency * a * t), sin(frequency * a * t))), dot: left to right
            for j, fn in ((2 * a - 2, math.cos), (2 * a - 1, math.sin)):
                term = fn(frequency * a * t) * x[:, j]
                acc = term if acc is None else acc + term
        return acc


def compose_f(leaves, xs, t):
    """Composed f, Model.scala:122-128: model1.f(ls, t) + model2.f(rs, t) on the left-nested tree ((a |+| b) |+| c)."""
    g = leaves[0].f(xs[0], t)
    for leaf, x in zip(leaves[1:], xs[1:]):
        g = g + leaf.f(x, t)
    return g


def seq_sum(v):
    """Seq.sum / foldLeft(0.0)(_ + _): the sequential fp64 sum (numpy's own `sum` adds pairwise; `cumsum` adds left to right)."""
    return float(np.cumsum(v)[-1])


def systematic_resampling(w1, u, tie_last=True):
    """Resampling.systematicResampling, Resampling.scala:63-72, with treeEcdf :52-58, normalise :21-24, findAllInTreeMap :36-46.
    tie_last = False: the canonical lower bound instead of the TreeMap's duplicate-key rule (the build's
    deviation D3) -- the only line of this file that is not the reference's behaviour, kept so that the build's choice can be held against
    the same independent arithmetic."""
    n = len(w1)
    total = seq_sum(w1)                                        # prob.foldLeft(0.0)(_ + _)
    normalised = w1 / total                                    # prob.map(x => x / total)
    ecdf_keys = np.cumsum(normalised)                          # scanLeft(0.0)(_ + _).drop(1)
    ks = (u + np.arange(n, dtype=np.float64)) / n              # Vector.range(0, n).map(i => (u + i) / n)
    first = np.searchsorted(ecdf_keys, ks, side="left")        # remMap.from(k).head: the smallest key >= k
    if np.any(first >= n):
        raise LookupError("findAllInTreeMap: `.head` of an empty map (a grid point above the last cumulative weight)")
    # TreeMap keys are unique: `tree ++ keys.zip(items)` overwrites an equal key, so the key maps to the LAST particle that has it
    if not tie_last:
        return first.astype(np.int64)
    last = np.searchsorted(ecdf_keys, ecdf_keys[first], side="right") - 1
    return last.astype(np.int64)


def stratified_resampling(w1, us, tie_last=True):
    """Resampling.stratifiedResampling, Resampling.scala:78-86: one uniform per slot, ks_i = (i + u_i) / n; the same TreeMap walk."""
    n = len(w1)
    ecdf_keys = np.cumsum(w1 / seq_sum(w1))                    # treeEcdf, :52-58
    ks = (np.arange(n, dtype=np.float64) + us) / n             # Vector.range(0, n).map(i => (i + scala.util.Random.nextDouble) / n)
    first = np.searchsorted(ecdf_keys, ks, side="left")
    if np.any(first >= n):
        raise LookupError("findAllInTreeMap: `.head` of an empty map")
    if not tie_last:
        return first.astype(np.int64)
    return (np.searchsorted(ecdf_keys, ecdf_keys[first], side="right") - 1).astype(np.int64)


def multinomial_resampling(w1, us):
    """Resampling.multinomialResampling, Resampling.scala:92-96: particles.size independent draws of breeze's Multinomial(weights) --
    third-party (breeze 0.13.x, build.sbt; not under /root/reference), its published `draw`: prob = uniform * sum, then the weights are
    subtracted one after the other and the first index at which prob <= 0 is returned.  Stated literally (the sequential differences,
    one row per draw); output in draw order."""
    total = seq_sum(w1)                                        # Multinomial's `sum`
    out = np.zeros(len(us), dtype=np.int64)
    for i, u in enumerate(us):
        rem = np.subtract.accumulate(np.concatenate(([u * total], w1)))[1:]   # ((prob - w_0) - w_1) - ...
        hit = np.nonzero(rem <= 0)[0]
        out[i] = hit[0] if len(hit) else 0                     # (`params.activeKeysIterator.next`: rounding left prob > 0 at the end)
    return out


def weigh_and_resample(w, ll, u, tie_last=True, resampler="systematic"):
    """stepFilter's tail, ParticleFilter.scala:124-128 (and :218-223): max, w1, ll, ess, ancestors.  u: the one uniform of systematic
    resampling, or the n per-slot uniforms of the other two."""
    mx = float(np.max(w))
    w1 = np.exp(w - mx)
    if resampler == "systematic":
        anc = systematic_resampling(w1, u, tie_last)
    elif resampler == "stratified":
        anc = stratified_resampling(w1, u, tie_last)
    else:
        anc = multinomial_resampling(w1, u)
    ll = ll + mx + math.log(seq_sum(w1) / len(w1))             # s.ll + max + log(ParticleFilter.mean(w1)), :522-524
    nw = w1 / seq_sum(w1)                                      # Resampling.normalise
    ess = int(math.floor(1 / seq_sum(nw * nw)))                # :431-434
    return anc, ll, ess


def poisson_log_probability(gamma, y):
    """PoissonModel.dataLikelihood, Model.scala:269,273: breeze Poisson(exp(gamma)).logProbabilityOf(y.toInt)
    = -lambda + k log(lambda) - lgamma(k + 1)."""
    lam = np.exp(gamma)
    k = int(y)
    return -lam + k * np.log(lam) - gammaln(k + 1.0)


def data_likelihood(kind, eta, y, scale, df):
    """dataLikelihood of the LEFTMOST model of the composition (Model.scala:118-120,132), as the case classes write it, in numpy / scipy:
    Poisson :269,273 -- Gaussian (LinearModel :252-258, SeasonalModel :227-233: the second argument of breeze's Gaussian is the standard
    deviation exp(scale)) -- Student's t :155-160 (the LOG-pdf multiplied by 1/v: as written) -- negative binomial :186-195 -- zero-inflated
    Poisson :298-307 -- Bernoulli :318-336 (link clamped at +-6, -1e99 floors) -- Beta(link, 1) :349-352 with link = exp(-gamma) :345."""
    from scipy import stats
    if kind == "poisson":
        return poisson_log_probability(eta, y)
    if kind in ("linear", "seasonal"):
        v = math.exp(scale)
        return stats.norm.logpdf(y, loc=eta, scale=v)                     # Gaussian(x, v).logPdf(y)
    if kind == "studentt":
        v = math.exp(scale)
        return 1 / v * stats.t.logpdf((y - eta) / v, df)                  # 1/v * StudentsT(df).logPdf((y - eta) / v)
    if kind == "negbin":
        size, mu, k = math.exp(scale), np.exp(eta), int(y)
        return (gammaln(size + k) - gammaln(k + 1.0) - gammaln(size) + size * np.log(size / (mu + size)) + k * np.log(mu / (mu + size)))
    if kind == "zip":
        p, k = math.exp(scale) / (1 + math.exp(scale)), int(y)
        if k == 0:
            return np.log(p + (1 - p) * np.exp(-np.exp(eta)))
        return -math.log(1 + math.exp(scale)) + k * eta - np.exp(eta) - gammaln(k + 1.0)
    if kind == "bernoulli":
        link = np.where(eta > 6, 1.0, np.where(eta < -6, 0.0, 1.0 / (1 + np.exp(-eta))))
        with np.errstate(divide="ignore"):
            if y == 1.0:
                return np.where(link == 0.0, -1e99, np.log(link))
            return np.where(link == 1.0, -1e99, np.log(1 - link))
    if kind == "beta":
        return stats.beta.logpdf(y, np.exp(-eta), 1.0)                    # new Beta(link(gamma), 1.0).logPdf(y)
    raise ValueError(kind)


def literal_filter(leaves, lgcp_precision, t, y, has, n, normals, uniform, tie_last=True, obs=("poisson", None, 0), resampler="systematic"):
    """llFilter, ParticleFilter.scala:137-140 over stepFilter :116-132 (FilterLgcp.stepFilter :210-226 when lgcp_precision > 0).
    normals(step, init, sub) -> [n, d] standard normals, uniform(step) -> u: the injected variates.  Returns per-observation ll and ess,
    the ancestors of every weighted observation and the final cloud."""
    dims = [lf.dim for lf in leaves]
    cuts = np.cumsum([0] + dims)
    split = lambda z: [z[:, cuts[i]:cuts[i + 1]] for i in range(len(leaves))]
    t0 = float(np.min(t))                                       # data.minBy(_.t).t
    xs = [lf.initial(z) for lf, z in zip(leaves, split(normals(0, True, -1)))]   # initialiseState, :105-108
    ll, ess, now = 0.0, n, t0
    ll_t, ess_t, ancs = [], [], []
    for s in range(len(t)):
        dt = float(t[s]) - now
        if lgcp_precision > 0:
            # calcWeight, :184-208: n = ceil(dt / 10^-precision) transitions of 10^-precision each, the clock started at y.t (:194,215)
            if dt == 0:
                g = compose_f(leaves, xs, float(t[s]))
                w = g - g
            else:
                delta = math.pow(10, -lgcp_precision)
                nsub = int(math.ceil(dt / delta))
                tau, haz = float(t[s]), np.zeros(n)
                for k in range(nsub):
                    xs = [lf.step(x, delta, z) for lf, x, z in zip(leaves, xs, split(normals(s, False, k)))]
                    tau = tau + delta
                    haz = haz + np.exp(compose_f(leaves, xs, tau)) * delta
                w = compose_f(leaves, xs, float(t[s])) - haz
            weighted = True
        else:
            xs = [lf.step(x, dt, z) for lf, x, z in zip(leaves, xs, split(normals(s, False, -1)))]   # :118
            weighted = bool(has[s])
            if weighted:
                w = data_likelihood(obs[0], compose_f(leaves, xs, float(t[s])), float(y[s]), obs[1], obs[2])   # :123
        if weighted:
            anc, ll, ess = weigh_and_resample(w, ll, uniform(s), tie_last, resampler)
            xs = [x[anc] for x in xs]                                                                 # :130
            ancs.append(anc)
        now = float(t[s])
        ll_t.append(ll); ess_t.append(ess)
    return np.array(ll_t), np.array(ess_t, dtype=np.int64), ancs, np.concatenate(xs, axis=1)


# ======================================================================================================= harness (not the witness)
def leaves_of(model):
    """The host mirror's parameterised model (composablestatespacemodels_amd.model.Model) -> Leaf objects, from the STORED parameters."""
    out = []
    for spec, node, sde in model.leaves:
        p = sde.params
        stored = {"m0": p.m0, "c0": p.c0}
        for name in ("mu", "phi", "sigma", "a", "b", "g"):
            if hasattr(p, name):
                stored[name] = getattr(p, name)
        out.append(Leaf(sde.kind, sde.dimension, stored, (spec.period, spec.harmonics) if spec.obs == "seasonal" else None))
    return out


CASES = [   # (name, N, T, missing fraction): BASELINE configs[0] at its own size, configs[1..3] at sizes the witness runs in seconds ...
    ("c1", 1000, 100, 0.0),
    ("c2", 4096, 60, 0.15),
    ("c3", 2048, 40, 0.0),
    ("c4", 2048, 24, 0.0),
    # ... and every other observation model and transition of SURVEY.md 8f-1 / A3 (tests/cases.py: the golden cases' models)
    ("linear", 2048, 40, 0.0), ("negbin", 2048, 40, 0.1), ("zip", 2048, 40, 0.0), ("bernoulli", 2048, 40, 0.0), ("studentt", 2048, 40, 0.0),
    ("beta", 2048, 40, 0.0), ("gbsg", 2048, 40, 0.0), ("euler", 2048, 40, 0.1),
]


RESAMPLER_CASES = [   # (name, N, T, missing fraction, resampler): the filter constructed with another `Resample[A]`
    ("c1", 1000, 60, 0.0, "stratified"), ("c2", 2048, 40, 0.1, "stratified"),
    ("c1", 1000, 40, 0.0, "multinomial"), ("c2", 1024, 30, 0.1, "multinomial"),
]


def run_case(name, n, T, missing, resampler="systematic"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from oracle import oracle
    model, t, y, has = cases.literal_case(name, T, missing=missing)
    spec0, node0 = model.leaves[0][0], model.leaves[0][1]
    obs = (spec0.obs, node0.scale, spec0.df)
    prec = 2 if name == "c4" else 0
    flags = oracle.LITERAL_SUMS | oracle.LIBM | oracle.TIE_LAST
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED, flags)          # (only its variate dump is used here)
    lib = oracle.lib()
    out = {"name": name, "n": n, "T": T, "missing": missing, "lgcp_precision": prec, "seed": cases.SEED, "resampler": resampler}

    def uniforms(s):
        if resampler == "systematic":
            return float(lib.oracle_c_u(cases.SEED, s))
        us = np.zeros(n)
        (lib.oracle_c_strat_u_v if resampler == "stratified" else lib.oracle_c_multi_u_v)(cases.SEED, s, n, us.ctypes.data_as(C.POINTER(C.c_double)))
        return us
    for key, tie_last in (("tie_last", True), ("tie_first", False)):
        ll_t, ess_t, ancs, cloud = literal_filter(leaves_of(model), prec, t, y, has, n,
                                                  lambda s, init, sub: o.dump_normals(s, init, sub), uniforms, tie_last, obs, resampler)
        out[key] = {"ll_t": [float(v).hex() for v in ll_t], "ess_t": [int(v) for v in ess_t],
                    "anc_first": [int(v) for v in ancs[0]], "anc_last": [int(v) for v in ancs[-1]],
                    "weighted": int(len(ancs)), "x0_last": [float(v).hex() for v in cloud[:, 0]]}
    return out


PMMH_CASE = {"name": "c2", "n": 512, "T": 25, "missing": 0.1, "delta": 0.05, "iters": 12, "seed": 7}     # (accepts and rejections)


def literal_mh(theta0, delta, iters, loglik, proposal_normals, accept_uniform):
    """MetropolisHastings.mhStep folded `iters` times from init (PMMH.scala:68-81, :121 ll = -1e99), with the proposal of
    Parameters.perturb(delta) (Parameters.scala:65-67: Gaussian(theta_k, sqrt(delta)) on every stored scalar), a symmetric transition
    and a flat prior (both log terms 0, as the reference's examples run it).  loglik(it, theta) -> the pseudo-marginal log-likelihood of
    the filter run for iteration it; proposal_normals(it) / accept_uniform(it): the injected variates."""
    cur, cur_ll, accepted = np.array(theta0, dtype=np.float64), -1e99, 0
    sd = math.sqrt(delta)
    ll, theta, acc = [], [], []
    for it in range(iters):
        prop = cur + sd * proposal_normals(it)                     # proposal(s.params)
        pll = loglik(it, prop)                                      # state = pf(propParams)
        a = pll + 0.0 + 0.0 - 0.0 - cur_ll - 0.0                    # :71-72
        if math.log(accept_uniform(it)) < a:                        # :74-75
            cur, cur_ll, accepted = prop, pll, accepted + 1
        ll.append(cur_ll); theta.append(cur.copy()); acc.append(accepted)
    return np.array(ll), np.array(theta), np.array(acc, dtype=np.int64)


def run_pmmh(case):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from oracle import oracle
    lib = oracle.lib()
    unparam, init = cases.c2_unparam(), cases.c2_params()
    t, y, has = cases.poisson_counts(case["T"], missing=case["missing"])
    theta0 = np.array(init.flattenParams())
    n, seed = case["n"], case["seed"]
    dp = C.POINTER(C.c_double)

    def normals(it):
        z = np.zeros(len(theta0)); lib.oracle_c_mh_normals(seed, it, len(theta0), z.ctypes.data_as(dp)); return z
    out = dict(case)
    for key, tie_last in (("tie_last", True), ("tie_first", False)):
        def loglik(it, theta):
            model = unparam.run(init.withFlat(theta))
            k = int(lib.oracle_c_derive_key(seed, it + 1))          # the key of filter run it + 1 (only the variates' addresses)
            o = oracle.OraclePf(model.descriptor(0), n, k, oracle.LITERAL_SUMS | oracle.LIBM | oracle.TIE_LAST)
            ll_t, _, _, _ = literal_filter(leaves_of(model), 0, t, y, has, n, lambda s, i, sub: o.dump_normals(s, i, sub),
                                           lambda s: float(lib.oracle_c_u(k, s)), tie_last)
            return float(ll_t[-1])
        ll, theta, acc = literal_mh(theta0, case["delta"], case["iters"], loglik, normals, lambda it: float(lib.oracle_c_mh_u(seed, it)))
        out[key] = {"ll": [float(v).hex() for v in ll], "accepted": [int(v) for v in acc], "theta": [[float(v).hex() for v in row] for row in theta]}
    return out


def main():
    runs = [run_case(*c) for c in CASES] + [run_case(*c) for c in RESAMPLER_CASES]
    out = os.path.join(HERE, "literal_runs.json")
    json.dump({"made_by": "tests/golden/make_literal.py (numpy " + np.__version__ + "): the numpy statement of the path on variates dumped from "
                          "the oracle's literal mode; doubles as C99 hex strings", "runs": runs, "pmmh": run_pmmh(PMMH_CASE)}, open(out, "w"), indent=0)
    for r in runs:
        print(r["name"], r["resampler"], "N", r["n"], "T", r["T"], "ll (TreeMap ties)", float.fromhex(r["tie_last"]["ll_t"][-1]), "ll (first key wins)", float.fromhex(r["tie_first"]["ll_t"][-1]),
              "first-observation ancestors that differ between the two:", int(np.sum(np.array(r["tie_last"]["anc_first"]) != np.array(r["tie_first"]["anc_first"]))))


    pm = json.load(open(out))["pmmh"]
    print("pmmh", {k: pm[k] for k in ("name", "n", "T", "delta", "iters")}, "accepted (TreeMap ties)", pm["tie_last"]["accepted"], "(first key wins)", pm["tie_first"]["accepted"])


if __name__ == "__main__":
    main()

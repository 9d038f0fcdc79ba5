"""The BASELINE.json configurations made concrete and the synthetic data they run on.

Data are produced by this repo's own simulator (numpy PCG64, seed 20260101): the reference's
SimulateData is out of scope and nothing here reads /root/reference.
"""
from __future__ import annotations

import numpy as np

from composablestatespacemodels_amd import Model, Parameters, Sde, SdeParameter

SEED = 20260101


def c1_model():
    """C1: Model.poisson(Sde.brownianMotion(1)), brownianParameter(0)(1)(0.01)."""
    p = Parameters.apply(None, SdeParameter.brownianParameter(0.0, 1.0, 0.01))
    return Model.poisson(Sde.brownianMotion(1)).run(p)


def c2_unparam():
    return Model.poisson(Sde.ouProcess(1)) | Model.seasonal(24, 1, Sde.ouProcess(2))


def c2_params():
    return (Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, 0.0, 0.3))
            | Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, 1.0, 0.3)))


def c2_model():
    """C2: poisson(ouProcess(1)) |+| seasonal(24, 1, ouProcess(2)), d = 3."""
    return c2_unparam().run(c2_params())


def c3_model():
    """C3: poisson(brownianMotion(1)) |+| seasonal(24, 4, ouProcess(8)), d = 9."""
    p = (Parameters.apply(None, SdeParameter.brownianParameter(0.0, 1.0, 0.01))
         | Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, [-1.0, -1.0, 0.0, 0.0, 0.0, 0.0, -0.125, -0.125], 0.3)))
    return (Model.poisson(Sde.brownianMotion(1)) | Model.seasonal(24, 4, Sde.ouProcess(8))).run(p)


def c4_model():
    """C4: Model.lgcp(Sde.ouProcess(1)), ouParameter(0.1)(0.5)(0.4)(0.1)(0.5), precision 2."""
    p = Parameters.apply(None, SdeParameter.ouParameter(0.1, 0.5, 0.4, 0.1, 0.5))
    return Model.lgcp(Sde.ouProcess(1)).run(p)


def lgcp_seasonal_model():
    """Model.lgcp(Sde.ouProcess(1)) |+| Model.seasonal(24, 1, Sde.ouProcess(2)): a log-Gaussian Cox process whose intensity has a
    daily cycle -- f depends on time, so FilterLgcp evaluates it at every sub-step time (model/ParticleFilter.scala:193-205)."""
    p = (Parameters.apply(None, SdeParameter.ouParameter(0.1, 0.5, 0.4, 0.1, 0.5))
         | Parameters.apply(None, SdeParameter.ouParameter(0.0, 0.3, 0.2, [0.3, -0.2], 0.2)))
    return (Model.lgcp(Sde.ouProcess(1)) | Model.seasonal(24, 1, Sde.ouProcess(2))).run(p)


def linear_model(sigma=0.3, obs_sd=0.5, m0=0.5, c0=2.0):
    """Model.linear(Sde.brownianMotion(1)): Kalman-tractable (docs/particle_filter.md:8-18)."""
    p = Parameters.apply(np.log(obs_sd), SdeParameter.brownianParameter(m0, c0, sigma))
    return Model.linear(Sde.brownianMotion(1)).run(p)


def gen_brownian_seasonal_gaussian():
    """A Gaussian-observation composition exercising GenBrownian + a seasonal leaf + scale."""
    p = (Parameters.apply(np.log(0.7), SdeParameter.genBrownianParameter(0.2, 1.5, 0.05, 0.2))
         | Parameters.apply(None, SdeParameter.ouParameter([0.0, 0.1], 1.0, 0.3, [0.5, -0.5], 0.2)))
    return (Model.linear(Sde.genBrownianMotion(1)) | Model.seasonal(12, 1, Sde.ouProcess(2))).run(p)


def euler_model():
    """A user-defined Sde on the trait's default Euler-Maruyama step (Sde.scala:23-43)."""
    p = Parameters.apply(None, SdeParameter.eulerAffineParameter([0.0, 0.5], 1.0, [-0.3, -0.1], [0.1, 0.0], [0.4, 0.2]))
    return Model.poisson(Sde.eulerAffine(2)).run(p)


def negbin_model():
    """Model.negativeBinomial(Sde.ouProcess(1)) |+| seasonal: mirrors examples/Simulation.scala:15-28."""
    p = (Parameters.apply(np.log(3.0), SdeParameter.brownianParameter(0.0, 1.0, 0.01))
         | Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, [0.25, -0.25], 0.3)))
    return (Model.negativeBinomial(Sde.brownianMotion(1)) | Model.seasonal(24, 1, Sde.ouProcess(2))).run(p)


def zip_model():
    p = Parameters.apply(-0.8, SdeParameter.ouParameter(0.5, 1.0, 0.2, 0.5, 0.3))
    return Model.zeroInflatedPoisson(Sde.ouProcess(1)).run(p)


def bernoulli_model():
    p = Parameters.apply(None, SdeParameter.ouParameter(0.0, 4.0, 0.2, 0.0, 1.5))
    return Model.bernoulli(Sde.ouProcess(1)).run(p)


def studentt_model():
    p = Parameters.apply(np.log(0.6), SdeParameter.genBrownianParameter(0.5, 2.0, 0.02, 0.3))
    return Model.studentsT(Sde.genBrownianMotion(1), 5).run(p)


def beta_model():
    p = Parameters.apply(None, SdeParameter.ouParameter(0.5, 0.2, 0.2, 0.5, 0.2))
    return Model.beta(Sde.ouProcess(1)).run(p)


def binary_series(T, seed=SEED):
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)
    return t, (rng.random(T) < 0.45).astype(np.float64), np.ones(T, dtype=np.uint8)


def unit_interval_series(T, seed=SEED):
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)
    return t, rng.beta(0.7, 1.0, T), np.ones(T, dtype=np.uint8)


def counts_with_zeros(T, seed=SEED):
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64)
    y = rng.poisson(1.5, T).astype(np.float64)
    y[rng.random(T) < 0.4] = 0.0
    return t, y, np.ones(T, dtype=np.uint8)


def poisson_counts(T, seed=SEED, rate=2.0, dt=1.0, missing=0.0):
    """Regular observation times 0, dt, 2dt ... with Poisson counts around a slowly varying rate."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64) * dt
    lam = rate * np.exp(0.5 * np.sin(2 * np.pi * t / 24.0) + 0.1 * rng.standard_normal(T).cumsum() / np.sqrt(np.arange(1, T + 1)))
    y = rng.poisson(lam).astype(np.float64)
    has = (rng.random(T) >= missing).astype(np.uint8)
    has[0] = 1
    return t, y, has


def gaussian_series(T, seed=SEED, dt=1.0, sigma=0.3, obs_sd=0.5):
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float64) * dt
    x = 0.5 + np.cumsum(np.sqrt(sigma * dt) * rng.standard_normal(T))
    y = x + obs_sd * rng.standard_normal(T)
    return t, y, np.ones(T, dtype=np.uint8)


def event_times(T, seed=SEED, horizon=10.0):
    """Irregular event times on [0, horizon] (what thinning a LGCP yields), first datum at t0."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.random(T) * horizon)
    t = np.round(t, 3)
    return t, np.ones(T), np.ones(T, dtype=np.uint8)


GOLDEN_NAMES = ["c1", "c2", "c3", "c4", "linear", "negbin", "zip", "bernoulli", "studentt", "beta"]


def golden_case(name, T, missing=0.0):
    """(model, t, y, has) of a committed run in tests/golden/oracle_runs.json."""
    mk = {"c1": c1_model, "c2": c2_model, "c3": c3_model, "c4": c4_model, "linear": linear_model, "negbin": negbin_model,
          "zip": zip_model, "bernoulli": bernoulli_model, "studentt": studentt_model, "beta": beta_model}[name]
    if name == "c4":
        data = event_times(T)
    elif name in ("linear", "studentt"):
        data = gaussian_series(T)
    elif name == "zip":
        data = counts_with_zeros(T)
    elif name == "bernoulli":
        data = binary_series(T)
    elif name == "beta":
        data = unit_interval_series(T)
    else:
        data = poisson_counts(T, missing=missing)
    return (mk(),) + tuple(data)


def literal_case(name, T, missing=0.0):
    """(model, t, y, has) of a run of the numpy witness (tests/golden/literal_runs.json): the golden cases plus the two compositions that
    exercise the generalised Brownian motion and the trait's Euler-Maruyama step."""
    if name == "gbsg":
        return (gen_brownian_seasonal_gaussian(),) + tuple(gaussian_series(T))
    if name == "euler":
        return (euler_model(),) + tuple(poisson_counts(T, missing=missing))
    return golden_case(name, T, missing=missing)


def max_dim_model():
    """d = CSSM_MAX_DIM = 16 in one composition: poisson(brownian 2) |+| seasonal(12, 3, ou 6) |+| linear-leaf(genBrownian 8)."""
    p = (Parameters.apply(None, SdeParameter.brownianParameter([0.0, 0.1], 1.0, [0.01, 0.02]))
         | Parameters.apply(None, SdeParameter.ouParameter(0.0, 0.5, 0.2, [0.1, -0.1, 0.0], 0.2))
         | Parameters.apply(0.0, SdeParameter.genBrownianParameter(0.0, 0.3, [0.01, -0.01], 0.05)))
    return (Model.poisson(Sde.brownianMotion(2)) | Model.seasonal(12, 3, Sde.ouProcess(6)) | Model.linear(Sde.genBrownianMotion(8))).run(p)


def many_leaves_model(n_leaves=16):
    """CSSM_MAX_LEAVES one-dimensional leaves."""
    ps = Parameters.apply(None, SdeParameter.ouParameter(0.0, 0.3, 0.2, 0.05, 0.1))
    um = Model.poisson(Sde.ouProcess(1))
    for i in range(n_leaves - 1):
        ps = ps | Parameters.apply(None, SdeParameter.brownianParameter(0.0, 0.05, 0.001 * (i + 1)))
        um = um | Model.poisson(Sde.brownianMotion(1))
    return um.run(ps)


def dim_model(d: int):
    """A model of total latent dimension d (1..16): poisson(ou(1 or 2)) |+| seasonal(24, h, ou(2 h)) -- one kernel
    instantiation (particles per thread, LDS staging layout) per dimension."""
    first = 1 if d % 2 else 2
    p = Parameters.apply(None, SdeParameter.ouParameter(0.0, 0.5, 0.2, [0.1, -0.05][:first], 0.2))
    m = Model.poisson(Sde.ouProcess(first))
    h = (d - first) // 2
    if h:
        p = p | Parameters.apply(None, SdeParameter.ouParameter(0.0, 0.2, 0.2, [0.05 * ((i % 3) - 1) for i in range(2 * h)], 0.1))
        m = m | Model.seasonal(24, h, Sde.ouProcess(2 * h))
    return m.run(p)

"""Pins of the CPU restatement (oracle/): the reference's own tests pin no filter number
(SURVEY.md 8c), so the oracle is pinned here by scipy golden vectors (K1), closed-form transition
moments and the double-logistic quirk (K2), hand-worked systematic-resampling examples that follow
Resampling.scala:52-72 literally (K3), the exact Kalman log-likelihood (K4), committed oracle runs
(K5) and the reference's own resampling property (SamplingTest.scala:16-18)."""
import json
import math
import os

import numpy as np
import pytest

import cases
from composablestatespacemodels_amd import Model, Parameters, Sde, SdeParameter, logistic
from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ----------------------------------------------------------------------------- K1 densities
def test_k1_poisson_density_matches_scipy_golden():
    g = json.load(open(os.path.join(GOLD, "densities.json")))
    f = oracle.lib().oracle_logdens_poisson
    for r in g["poisson"]:
        got = f(r["gamma"], r["y"])
        assert abs(got - r["logpmf"]) <= 2e-12 * max(1.0, abs(r["logpmf"])), r


def test_k1_gaussian_density_matches_scipy_golden():
    g = json.load(open(os.path.join(GOLD, "densities.json")))
    f = oracle.lib().oracle_logdens_gaussian
    for r in g["gaussian"]:
        got = f(r["gamma"], r["y"], r["sd"])
        assert abs(got - r["logpdf"]) <= 1e-12 * max(1.0, abs(r["logpdf"])), r


def test_k1_next_row_densities_match_scipy_golden():
    """NegBin, zero-inflated Poisson, Bernoulli, Student-t (with the reference's 1/v factor on the
    LOG-pdf, Model.scala:158) and Beta(exp(-gamma), 1): Model.scala:155-160,186-195,298-307,318-336,349-352."""
    g = json.load(open(os.path.join(GOLD, "densities.json")))
    L = oracle.lib()
    tol = lambda ref: 2e-12 * max(1.0, abs(ref))
    for r in g["negbin"]:
        assert abs(L.oracle_logdens_negbin(r["gamma"], r["y"], r["size"]) - r["logpmf"]) <= tol(r["logpmf"]), r
    for r in g["zip"]:
        assert abs(L.oracle_logdens_zip(r["gamma"], r["y"], r["v"]) - r["logpmf"]) <= tol(r["logpmf"]), r
    for r in g["bernoulli"]:
        assert abs(L.oracle_logdens_bernoulli(r["gamma"], r["y"]) - r["logp"]) <= tol(r["logp"]), r
    for r in g["student_t"]:
        assert abs(L.oracle_logdens_student_t(r["gamma"], r["y"], r["v"], r["df"]) - r["val"]) <= tol(r["val"]), r
    for r in g["beta"]:
        assert abs(L.oracle_logdens_beta(r["gamma"], r["y"]) - r["logpdf"]) <= tol(r["logpdf"]), r
    # the Bernoulli link's clamp and floor, Model.scala:318-336
    assert L.oracle_logdens_bernoulli(6.5, 0.0) == -1e99 and L.oracle_logdens_bernoulli(-6.5, 1.0) == -1e99
    assert L.oracle_logdens_bernoulli(6.5, 1.0) == 0.0 and L.oracle_logdens_bernoulli(-6.5, 0.0) == 0.0


def test_lgamma_real_argument():
    from scipy import special
    for x in [1e-3, 0.3, 0.5, 1.0, 1.5, 2.0, 2.94, 7.25, 23.9, 24.0, 100.5, 1e4, 1e6]:
        ref = float(special.gammaln(x))
        assert abs(oracle.lib().oracle_c_lgamma(x) - ref) <= 3e-14 * max(1.0, abs(ref))


# ----------------------------------------------------------------------------- K2 transitions
def test_k2_double_logistic_on_ou_phi():
    # ouParameter(...)(0.2) stores logistic(0.2); OuProcess applies logistic again (Sde.scala:136)
    p = Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, 0.5, 0.3))
    m = Model.poisson(Sde.ouProcess(1)).run(p)
    comp = oracle.OraclePf(m.descriptor(), 8, 1).components()[0]
    assert abs(comp[3] - logistic(logistic(0.2))) < 1e-15
    assert abs(comp[3] - 0.634) < 1e-3
    assert abs(comp[1] - 1.0) < 1e-15 and abs(comp[4] - 0.3) < 1e-15 and comp[2] == 0.5


def test_k2_cyclic_parameter_repeat():
    # Sde.buildParamRepeat (Sde.scala:177-179)
    p = Parameters.apply(None, SdeParameter.ouParameter([0.0, 1.0], 1.0, 0.2, [-1.0, -1.0, 0.0], 0.3))
    m = Model.poisson(Sde.ouProcess(5)).run(p)
    c = oracle.OraclePf(m.descriptor(), 4, 1).components()
    np.testing.assert_array_equal(c[:, 0], [0.0, 1.0, 0.0, 1.0, 0.0])
    np.testing.assert_array_equal(c[:, 2], [-1.0, -1.0, 0.0, -1.0, -1.0])


def _one_step_moments(model, x0, dt, n=400000):
    o = oracle.OraclePf(model.descriptor(), n, 99)
    o.init_from(0.0, x0)
    o.step(dt, None, has_obs=False)
    return o.particles()


def test_k2_brownian_is_variance_rate_and_dt0_is_identity():
    p = Parameters.apply(None, SdeParameter.brownianParameter(0.0, 1.0, 0.25))
    m = Model.poisson(Sde.brownianMotion(1)).run(p)
    x = _one_step_moments(m, [1.5], 2.0)[0]
    assert abs(x.mean() - 1.5) < 4e-3 and abs(x.var() - 0.25 * 2.0) < 6e-3    # N(x, sigma*dt), docs/model.md:17
    x = _one_step_moments(m, [1.5], 0.0, n=1000)[0]
    np.testing.assert_array_equal(x, np.full(1000, 1.5))


def test_k2_gen_brownian_and_ou_moments():
    p = Parameters.apply(None, SdeParameter.genBrownianParameter(0.0, 1.0, 0.3, 0.5))
    m = Model.poisson(Sde.genBrownianMotion(1)).run(p)
    x = _one_step_moments(m, [-1.0], 1.5)[0]
    assert abs(x.mean() - (-1.0 + 0.3 * 1.5)) < 5e-3 and abs(x.var() - 0.5 * 1.5) < 8e-3
    p = Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, 2.0, 0.7))
    m = Model.poisson(Sde.ouProcess(1)).run(p)
    phi = logistic(logistic(0.2)); dt = 0.8
    x = _one_step_moments(m, [0.5], dt)[0]
    assert abs(x.mean() - (2.0 + (0.5 - 2.0) * math.exp(-phi * dt))) < 4e-3
    assert abs(x.var() - 0.7**2 / (2 * phi) * (1 - math.exp(-2 * phi * dt))) < 4e-3


def test_k2_euler_maruyama_default_step():
    m = cases.euler_model()
    x = _one_step_moments(m, [1.0, -2.0], 0.5)
    # x + (a + b x) dt + g sqrt(dt) z, Sde.scala:36-43
    assert abs(x[0].mean() - (1.0 + (0.1 - 0.3 * 1.0) * 0.5)) < 3e-3 and abs(x[0].var() - 0.4**2 * 0.5) < 2e-3
    assert abs(x[1].mean() - (-2.0 + (0.0 - 0.1 * -2.0) * 0.5)) < 3e-3 and abs(x[1].var() - 0.2**2 * 0.5) < 1e-3


def test_k2_initial_state_moments_and_composed_leaf_order():
    m = cases.c3_model()
    o = oracle.OraclePf(m.descriptor(), 200000, 5)
    o.init(0.0)
    x = o.particles()
    assert x.shape == (9, 200000)   # composed state size == sde.dimension (ModelTest.scala:72-86)
    assert np.all(np.abs(x.mean(axis=1)) < 0.01) and np.all(np.abs(x.var(axis=1) - 1.0) < 0.02)


# ----------------------------------------------------------------------------- K3 resampling
def test_k3_hand_worked_four_particles():
    # weights (unnormalised) -> normalise 0.1,0.2,0.3,0.4 -> ecdf .1,.3,.6,1.0; u = .5 -> ks .125,.375,.625,.875
    w = np.array([1.0, 2.0, 3.0, 4.0])
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.5, oracle.LITERAL_SUMS), [1, 2, 3, 3])
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.5), [1, 2, 3, 3])
    # u = 0.2 -> ks .05,.3,.55,.8 ; the grid point .3 falls ON the key .3 (TreeMap.from is inclusive) in exact
    # arithmetic; in fp64 the literal cumulative sum 0.1+0.2 = 0.30000000000000004 >= 0.3 as well
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.2, oracle.LITERAL_SUMS), [0, 1, 2, 3])


def test_k3_hand_worked_eight_particles_with_zero_weights_both_tie_policies():
    w = np.array([0.0, 0.25, 0.0, 0.0, 0.5, 0.0, 0.25, 0.0])
    # ecdf: 0,.25,.25,.25,.75,.75,1,1 ; u=.5: ks = (.5+i)/8 = .0625,.1875,.3125,...,.9375
    first = [1, 1, 4, 4, 4, 4, 6, 6]             # first key >= k (contract: lower bound)
    last = [3, 3, 5, 5, 5, 5, 7, 7]              # TreeMap keeps the LAST value per key (Resampling.scala:57)
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.5), first)
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.5, oracle.LITERAL_SUMS), first)
    np.testing.assert_array_equal(oracle.resample_systematic(w, 0.5, oracle.LITERAL_SUMS | oracle.TIE_LAST), last)


def test_k3_uniform_one_hot_counts_monotone_and_length():
    rng = np.random.default_rng(7)
    for n in (1, 2, 5, 64, 1000, 4097):
        # u == 0 puts the grid point i/n exactly ON the key C_{i-1}; TreeMap.from is inclusive
        # (Resampling.scala:41), so slot i takes particle i-1 -- literal reference semantics
        np.testing.assert_array_equal(oracle.resample_systematic(np.ones(n), 0.0), np.maximum(np.arange(n) - 1, 0))
        for u in (1e-9, 0.3, float(rng.random()), 1 - 2.0**-53):
            np.testing.assert_array_equal(oracle.resample_systematic(np.ones(n), u), np.arange(n))
            k = int(rng.integers(n))
            w = np.zeros(n); w[k] = 3.0
            np.testing.assert_array_equal(oracle.resample_systematic(w, u), np.full(n, k))
        w = rng.random(n)
        a = oracle.resample_systematic(w, float(rng.random()))
        assert len(a) == n                               # the reference's property, SamplingTest.scala:16-18
        assert np.all(np.diff(a.astype(np.int64)) >= 0)
        counts = np.bincount(a, minlength=n)
        expect = n * w / w.sum()
        assert np.all(counts >= np.floor(expect - 1e-9)) and np.all(counts <= np.ceil(expect + 1e-9))


def test_k3_contract_and_literal_sums_agree():
    """The fixed-point scan and the reference's sequential fp64 scan differ by <= N*2^-53: identical
    ancestors on dyadic weights, and at most a vanishing fraction of flipped slots on random ones."""
    rng = np.random.default_rng(8)
    w = rng.integers(0, 16, 4096) / 16.0
    w[0] = 1.0
    for u in (0.0, 0.37, 0.999):
        np.testing.assert_array_equal(oracle.resample_systematic(w, u), oracle.resample_systematic(w, u, oracle.LITERAL_SUMS))
    w = np.exp(-rng.exponential(5.0, 100000)); w[17] = 1.0
    a, Ca = oracle.resample_systematic(w, 0.61803, want_cumw=True)
    b, Cb = oracle.resample_systematic(w, 0.61803, oracle.LITERAL_SUMS, want_cumw=True)
    assert np.abs(Ca - Cb).max() < 100000 * 2.0**-52
    assert Ca[-1] == 1.0 and np.all(np.diff(Ca) >= 0)
    assert np.mean(a != b) < 1e-4


def test_k3_literal_mode_reproduces_the_empty_map_failure():
    # a grid point above the last cumulative weight makes `.head` throw in the reference (Resampling.scala:41-42)
    # for these weights the sequential fp64 scan of w/total ends at 0.9999999999999998 while the last
    # grid point (u + n - 1)/n rounds to 1.0 for u = 1 - 2^-53
    w = np.array([1, 3, 5, 4, 4, 1, 1, 2, 1, 7, 5, 6, 3, 6, 7, 4, 5, 9, 8, 9, 4, 7, 9], dtype=np.float64)
    with pytest.raises(oracle.OracleError) as e:
        oracle.resample_systematic(w, 1 - 2.0**-53, oracle.LITERAL_SUMS | oracle.TIE_LAST)
    assert e.value.code == oracle.EEMPTY
    # the contract's cumulative weights end at exactly 1.0, so it cannot happen there
    a, C = oracle.resample_systematic(w, 1 - 2.0**-53, want_cumw=True)
    assert C[-1] == 1.0 and a[-1] == len(w) - 1


# ----------------------------------------------------------------------------- K4 Kalman
def kalman_ll(t, y, m0, c0, sigma, obs_sd):
    """Exact log-likelihood of Brownian latent (variance rate sigma) + Gaussian observation."""
    m, c, tp, ll = m0, c0, float(np.min(t)), 0.0
    for ts, ys in zip(t, y):
        c = c + sigma * (ts - tp); tp = ts
        s = c + obs_sd**2
        ll += -0.5 * (math.log(2 * math.pi * s) + (ys - m)**2 / s)
        k = c / s
        m, c = m + k * (ys - m), (1 - k) * c
    return ll


def test_k4_particle_filter_likelihood_converges_to_kalman():
    sigma, obs_sd, m0, c0 = 0.3, 0.5, 0.5, 2.0
    model = cases.linear_model(sigma, obs_sd, m0, c0)
    t, y, has = cases.gaussian_series(40, sigma=sigma, obs_sd=obs_sd)
    exact = kalman_ll(t, y, m0, c0, sigma, obs_sd)
    errs = []
    for n in (500, 50000):
        lls = [oracle.OraclePf(model.descriptor(), n, 1000 + r).filter(t, y, has)[0] for r in range(8)]
        errs.append(np.mean(np.abs(np.array(lls) - exact)))
    # measured: 0.22 at N=500, 0.026 at N=50000 (sd of the estimate 0.18 -> 0.03, i.e. ~ 1/sqrt(N))
    assert errs[1] < 0.06, (exact, errs)
    assert errs[1] < errs[0] / 4          # Monte-Carlo error shrinks with N


# ----------------------------------------------------------------------------- K5 committed oracle runs
@pytest.mark.parametrize("name", cases.GOLDEN_NAMES)
def test_k5_oracle_reproduces_committed_runs(name):
    g = json.load(open(os.path.join(GOLD, "oracle_runs.json")))[name]
    model, t, y, has = cases.golden_case(name, g["T"], g["missing"])
    o = oracle.OraclePf(model.descriptor(g["lgcp_precision"]), g["n"], cases.SEED)
    ll, ll_t, ess_t, path = o.filter(t, y, has, want_path=True)
    assert ll == float.fromhex(g["ll"])
    np.testing.assert_array_equal(ll_t, [float.fromhex(v) for v in g["ll_t"]])
    np.testing.assert_array_equal(ess_t, g["ess_t"])
    np.testing.assert_array_equal(o.ancestors(), g["ancestors_last"])
    np.testing.assert_array_equal(path, [[float.fromhex(v) for v in row] for row in g["path"]])


# ----------------------------------------------------------------------------- arithmetic modes
def test_contract_literal_and_libm_modes_agree_on_the_likelihood():
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(50, missing=0.1)
    n = 3000
    ll_c = oracle.OraclePf(model.descriptor(), n, 11).filter(t, y, has)[0]
    ll_l = oracle.OraclePf(model.descriptor(), n, 11, oracle.LITERAL_SUMS).filter(t, y, has)[0]
    assert abs(ll_c - ll_l) < 1e-9          # same particles unless an ancestor flips; sums differ by <= N*2^-53
    ll_m = oracle.OraclePf(model.descriptor(), 200, 11, oracle.LIBM | oracle.LITERAL_SUMS).filter(t[:8], y[:8], has[:8])[0]
    ll_s = oracle.OraclePf(model.descriptor(), 200, 11).filter(t[:8], y[:8], has[:8])[0]
    assert abs(ll_m - ll_s) < 1e-8          # glibc vs contract elementary functions: last-ulp differences only


def test_missing_observation_keeps_ll_and_ess():
    model = cases.c1_model()
    o = oracle.OraclePf(model.descriptor(), 500, 3)
    o.init(0.0)
    ll0, ess0 = o.step(1.0, 2.0)
    ll1, ess1 = o.step(2.0, None, has_obs=False)     # ParticleFilter.scala:121
    assert (ll1, ess1) == (ll0, ess0)
    np.testing.assert_array_equal(o.ancestors(), np.arange(500))


def test_lgcp_dt_zero_gives_zero_weights():
    model = cases.c4_model()
    o = oracle.OraclePf(model.descriptor(2), 256, 3)
    o.init(1.0)
    before = o.particles()
    ll, ess = o.step(1.0, 1.0)           # dt == 0: weight f - f = 0, state kept (ParticleFilter.scala:212-213)
    assert ll == 0.0 and ess == 256
    np.testing.assert_array_equal(o.proposed(), before)
    np.testing.assert_array_equal(o.logw(), np.zeros(256))


def test_nonfinite_weights_are_an_error():
    model = cases.linear_model()
    o = oracle.OraclePf(model.descriptor(), 64, 3)
    o.init(0.0)
    with pytest.raises(oracle.OracleError) as e:
        o.step(1.0, float("nan"))
    assert e.value.code == oracle.ENONFINITE


# ----------------------------------------------------------------------------- cloud summaries
def test_summary_follows_the_reference_index_conventions():
    """getCredibleInterval uses sorted(N - index - 1), sorted(index - 1); getOrderStatistic uses sorted(N - index),
    sorted(index) (ParticleFilter.scala:455-460,488-502) -- both off-by-one conventions are reproduced."""
    model = cases.c2_model()
    n = 1000
    o = oracle.OraclePf(model.descriptor(), n, 3)
    o.init(0.0)
    o.step(1.0, 2.0)
    x = o.particles()
    m, lo, hi, em, el, eu = o.summary(0.975)
    idx = int(np.floor(0.975 * n))
    for k in range(3):
        srt = np.sort(x[k])
        assert lo[k] == srt[n - idx - 1] and hi[k] == srt[idx - 1]
        assert abs(m[k] - x[k].mean()) < 1e-13
    t = 1.0
    eta = np.exp(x[0] + np.cos(2 * np.pi * t / 24) * x[1] + np.sin(2 * np.pi * t / 24) * x[2])
    srt = np.sort(eta)
    assert abs(el - srt[n - idx]) <= 1e-15 * abs(el) * 4 and abs(eu - srt[idx]) <= 1e-15 * abs(eu) * 4
    assert abs(em - np.exp(m[0] + np.cos(2 * np.pi * t / 24) * m[1] + np.sin(2 * np.pi * t / 24) * m[2])) < 1e-12


# ----------------------------------------------------------------------------- other resamplers
def test_stratified_and_multinomial_properties():
    rng = np.random.default_rng(21)
    n = 4000
    w = rng.random(n); w[rng.random(n) < 0.3] = 0.0
    a = oracle.resample_stratified(w, 5, 7)
    assert len(a) == n and np.all(np.diff(a.astype(np.int64)) >= 0)      # ordered grid -> monotone ancestors
    counts = np.bincount(a, minlength=n)
    expect = n * w / w.sum()
    assert np.all(counts[w == 0] == 0) and np.all(np.abs(counts - expect) < 2.0)   # stratified: |count - N w| < 2
    np.testing.assert_array_equal(oracle.resample_stratified(np.ones(64), 1), np.arange(64))
    m = oracle.resample_multinomial(w, 5, 7)
    assert len(m) == n and np.all(w[m] > 0)                                # zero-weight particles are never drawn
    assert not np.all(np.diff(m.astype(np.int64)) >= 0)                    # draw order, not sorted
    big = oracle.resample_multinomial(np.array([0.1, 0.2, 0.3, 0.4]), 3)
    cnt = np.bincount(oracle.resample_multinomial(np.tile([0.1, 0.2, 0.3, 0.4], 5000), 9) % 4, minlength=4) / 20000.0
    assert np.all(np.abs(cnt - [0.1, 0.2, 0.3, 0.4]) < 0.02) and len(big) == 4


def test_residual_resampling_extension_hand_worked_and_unbiased():
    """The EXTENSION (the resampler Resampling.scala:124-129 describes; its body cannot run), pinned by hand: w = (1/2, 1/4, 1/8, 1/8),
    n = 4 -> n w = (2, 1, 1/2, 1/2): particle 0 twice, particle 1 once, one slot left, drawn from the residuals (0, 0, 1/2, 1/2):
    particle 2 if its uniform is <= 1/2, else 3.  And its defining property over many seeds: E[count_i] = n w_i."""
    w = np.array([0.5, 0.25, 0.125, 0.125])
    seen = set()
    for seed in range(40):
        a = oracle.resample_residual(w, seed)
        assert list(a[:3]) == [0, 0, 1] and a[3] in (2, 3)
        seen.add(int(a[3]))
    assert seen == {2, 3}
    np.testing.assert_array_equal(oracle.resample_residual(np.ones(16), 3), np.arange(16))      # n w_i = 1: every particle once, no draw
    np.testing.assert_array_equal(oracle.resample_residual(np.array([0.0, 3.0, 0.0]), 3), [1, 1, 1])
    rng = np.random.default_rng(4)
    w = rng.random(50)
    cnt = np.zeros(50)
    for seed in range(400):
        cnt += np.bincount(oracle.resample_residual(w, seed), minlength=50)
    expect = 50 * w / w.sum()
    assert np.all(np.abs(cnt / 400 - expect) < 0.12)                        # unbiased: the residual part has variance < 1/4 per draw
    assert np.all(np.bincount(oracle.resample_residual(w, 1), minlength=50) >= np.floor(expect))


def test_filters_with_other_resamplers_estimate_the_same_likelihood():
    model = cases.linear_model()
    t, y, has = cases.gaussian_series(30)
    from test_oracle_pins import kalman_ll
    exact = kalman_ll(t, y, 0.5, 2.0, 0.3, 0.5)
    for flag in (oracle.RESAMPLE_STRATIFIED, oracle.RESAMPLE_MULTINOMIAL):
        lls = [oracle.OraclePf(model.descriptor(), 20000, 50 + r, flag).filter(t, y, has)[0] for r in range(4)]
        assert abs(np.mean(lls) - exact) < 0.1

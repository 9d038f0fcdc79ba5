"""PMMH host loop (model/PMMH.scala:68-81,114-123).

K6: with the filter stubbed to a deterministic function, the accept/reject sequence of mhStep for a
fixed uniform stream, including the -1e99 start.  The chain of the oracle's own restatement is pinned
for determinism on CPU; the native chain is compared with it on the GPU.
"""
import math

import numpy as np
import pytest

import cases
from composablestatespacemodels_amd import Parameters
from composablestatespacemodels_amd.pmmh import MetropolisHastings, MetropState, ParametersProposal
from oracle import oracle


class FixedStream:
    """A 'generator' that replays given normals and uniforms."""

    def __init__(self, normals, uniforms):
        self.z, self.u = list(normals), list(uniforms)

    def standard_normal(self, n):
        out = np.array(self.z[:n]); self.z = self.z[n:]
        return out

    def random(self):
        return self.u.pop(0)


def test_k6_mh_step_accept_reject_sequence_with_stub_filter():
    init = cases.c2_params()
    nt = len(init.flattenParams())
    lls = iter([-50.0, -60.0, -49.5, -49.9, -70.0])
    pf = lambda p: (next(lls), [np.zeros(3)])
    rng = FixedStream(np.zeros(nt * 5), [0.5, 0.5, 0.5, 0.7, 1e-12])
    chain = MetropolisHastings.pmmhState(init, ParametersProposal(0.05), lambda a, b: 0.0, lambda p: 0.0)(pf, rng, iters=5)
    out = [(s.ll, s.accepted) for s in chain]
    # 1: from -1e99 always accepted; 2: log(.5) < -10 false; 3: a=+0.5 accept; 4: log(.7)=-0.357 < -0.4 false
    # 5: log(1e-12)=-27.6 < -20.5 accept
    assert out == [(-50.0, 1), (-50.0, 1), (-49.5, 2), (-49.5, 2), (-70.0, 3)]


def test_approx_pmmh_reestimates_the_current_likelihood_every_step():
    """ApproxPMMH (PMMH.scala:128-152): two filter runs per iteration, proposal first; a rejected step keeps the fresh
    estimate of the current parameters, not the stored one."""
    init = cases.c2_params()
    nt = len(init.flattenParams())
    calls = []
    lls = iter([-50.0, -1e9,      # it 1: proposal -50, "current" (initial params) -1e9 -> accept
                -60.0, -50.5,     # it 2: proposal -60, current re-estimated -50.5: log(.5) < -9.5 false -> reject, ll becomes -50.5
                -49.0, -50.2])    # it 3: proposal -49, current -50.2: a = 1.2 -> accept
    def pf(p):
        calls.append(tuple(p.flattenParams()))
        return next(lls), [np.full(3, len(calls))]
    rng = FixedStream(np.ones(nt * 3), [0.5, 0.5, 0.5])
    chain = MetropolisHastings.approxPmmh(init, ParametersProposal(0.04), lambda a, b: 0.0, lambda p: 0.0)(pf, rng, iters=3)
    out = list(chain)
    assert [(s.ll, s.accepted) for s in out] == [(-50.0, 1), (-50.5, 1), (-49.0, 2)]
    assert len(calls) == 6 and calls[1] == tuple(init.flattenParams()) and calls[3] == calls[0]   # current = the accepted proposal
    assert out[1].sde[0] == 4                     # the rejected step carries the state of the re-run (4th filter call)
    ps = list(MetropolisHastings.params(iter(out)))
    assert [(q.ll, q.accepted) for q in ps] == [(-50.0, 1), (-50.5, 1), (-49.0, 2)] and ps[2].params is out[2].params


def test_pmmh_step_on_a_log_posterior_pair():
    """MetropolisHastings.pmmhStep (PMMH.scala:177-191)."""
    rng = FixedStream([0.0] * 3, [0.5, 0.5, 1e-9])
    pos = iter([-10.0, -12.0, -30.0])
    s = (-1e99, 0.0)
    out = []
    for _ in range(3):
        s = MetropolisHastings.pmmhStep(lambda p: next(pos), lambda p, r: p + 1.0 + r.standard_normal(1)[0], s, rng)
        out.append(s)
    # accept (from -1e99); log(.5) < -2 false -> stay; log(1e-9) = -20.7 < -20 -> accept
    assert out == [(-10.0, 1.0), (-10.0, 1.0), (-30.0, 2.0)]


def test_perturb_proposal_moments_and_shape():
    p = cases.c2_params()
    rng = np.random.default_rng(0)
    prop = ParametersProposal(0.04)
    draws = np.array([prop(p, rng).flattenParams() for _ in range(4000)])
    base = np.array(p.flattenParams())
    assert np.all(np.abs(draws.mean(axis=0) - base) < 0.02) and np.all(np.abs(draws.std(axis=0) - 0.2) < 0.02)


def test_oracle_pmmh_chain_is_deterministic_and_first_proposal_accepted():
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(12)
    theta0 = np.array(model.parameters().flattenParams())
    runs = []
    for _ in range(2):
        o = oracle.OraclePf(model.descriptor(), 300, 1)
        runs.append(o.pmmh(model.descriptor(), theta0, 0.01, t, y, has, seed=77, n_iters=6))
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)
    ll, th, acc, last = runs[0]
    assert acc[0] == 1 and np.all(np.diff(acc) >= 0) and ll[0] > -1e98
    assert not np.array_equal(th[0], theta0)           # the first proposal was adopted
    for i in range(1, 6):                              # a rejected step repeats the state
        if acc[i] == acc[i - 1]:
            assert ll[i] == ll[i - 1] and np.array_equal(th[i], th[i - 1])


@pytest.mark.gpu
def test_native_pmmh_chain_matches_oracle_bit_for_bit():
    from composablestatespacemodels_amd.pmmh import pmmh_native
    from composablestatespacemodels_amd import Data
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(15, missing=0.1)
    data = [Data(float(a), float(b) if h else None) for a, b, h in zip(t, y, has)]
    init = model.parameters()
    ll, th, acc, last = pmmh_native(cases.c2_unparam(), init, data, 1500, 0.01, 8, seed=99)
    o = oracle.OraclePf(model.descriptor(), 1500, 1)
    oll, oth, oacc, olast = o.pmmh(model.descriptor(), np.array(init.flattenParams()), 0.01, t, y, has, seed=99, n_iters=8)
    np.testing.assert_array_equal(ll, oll)
    np.testing.assert_array_equal(th, oth)
    np.testing.assert_array_equal(acc, oacc)
    np.testing.assert_array_equal(last, olast)


@pytest.mark.gpu
def test_native_pmmh_chain_at_the_full_c5_size_matches_oracle():
    """BASELINE config 5 at its full size -- N = 100 000 particles, T = 500 observations -- for the two iterations the
    one-thread oracle can afford (2 x 5e7 particle-steps): log-likelihoods, parameters, acceptances and the sampled states of
    the native chain equal the oracle's bit for bit."""
    from composablestatespacemodels_amd.pmmh import pmmh_native
    from composablestatespacemodels_amd import Data
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(500)
    data = [Data(float(a), float(b) if h else None) for a, b, h in zip(t, y, has)]
    init = model.parameters()
    ll, th, acc, last = pmmh_native(cases.c2_unparam(), init, data, 100_000, 0.05, 2, seed=7)
    o = oracle.OraclePf(model.descriptor(), 100_000, 1)
    oll, oth, oacc, olast = o.pmmh(model.descriptor(), np.array(init.flattenParams()), 0.05, t, y, has, seed=7, n_iters=2)
    np.testing.assert_array_equal(ll, oll)
    np.testing.assert_array_equal(th, oth)
    np.testing.assert_array_equal(acc, oacc)
    np.testing.assert_array_equal(last, olast)


@pytest.mark.gpu
def test_generic_pmmh_state_over_native_bootstrap_filter():
    from composablestatespacemodels_amd.pmmh import bootstrap_filter
    from composablestatespacemodels_amd import Data
    t, y, has = cases.poisson_counts(10)
    data = [Data(float(a), float(b)) for a, b in zip(t, y)]
    pf = bootstrap_filter(cases.c2_unparam(), data, 2000, seed=5)
    run = MetropolisHastings.pmmhState(cases.c2_params(), ParametersProposal(0.01), lambda a, b: 0.0, lambda p: 0.0)
    states = list(run(pf, np.random.default_rng(1), iters=5))
    assert len(states) == 5 and states[0].accepted == 1 and math.isfinite(states[-1].ll)
    assert states[-1].sde.shape == (3,)


def test_chains_with_adjacent_seeds_share_no_filter_key():
    """The Philox key of a chain's k-th filter run is a PRF of (seed, k) (cssm_pf_run_key), not seed + k: chains seeded s and
    s + 1 -- how two chains are naturally started side by side (examples/DetermineParameters.scala:68-69) -- must not replay
    each other's filter randomness one iteration apart.  The host function needs no GPU."""
    from composablestatespacemodels_amd import _abi
    lib = _abi.load_library()
    runs = range(1, 2001)
    for s in (0, 99, 20260101, 2**64 - 2):
        a = {lib.cssm_pf_run_key(s, k) for k in runs}
        b = {lib.cssm_pf_run_key((s + 1) % 2**64, k) for k in runs}
        assert len(a) == len(runs) and len(b) == len(runs)
        assert not (a & b)
        assert s not in a and (s + 1) % 2**64 not in a       # nor the handle's creation seed
    # the oracle's chain derives the same keys (shared header): the native-vs-oracle chain test relies on it
    from oracle import oracle
    assert oracle.lib() is not None

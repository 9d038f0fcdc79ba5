"""-m gpu: B independent filters per launch (include/cssm_pf.h: cssm_pfb_*, csrc/cssm_batch.hip) -- the chains of a PMMH run
(model/PMMH.scala:68-81,114-123; examples/DetermineParameters.scala:68-69 runs two side by side) and the parameter grid of a pilot run
(model/Streaming.scala:38-39) advanced in lockstep by one host thread.  Per chain the results must be those of a handle of its own, and
of the oracle: bit for bit."""
import numpy as np
import pytest

import cases
from composablestatespacemodels_amd.filter import NativePf, NativePfBatch
from oracle import oracle

pytestmark = pytest.mark.gpu


def _perturbed(make_params, k):
    """chain k's parameters: the model's own, shifted a little per chain (the structures stay equal)"""
    p = make_params()
    th = np.asarray(p.flattenParams())
    return p.withFlat(th + 0.03 * k * np.cos(np.arange(th.size) + k))


@pytest.mark.parametrize("n,B,whole", [(5000, 3, 0), (100_000, 4, 0), (70 * 1024 + 5, 2, 1), (3000, 5, 2), (1 << 20, 2, 0)])
def test_batched_filters_equal_single_handles_and_the_oracle(n, B, whole):
    um = cases.c2_unparam()
    models = [um.run(_perturbed(cases.c2_params, k)) for k in range(B)]
    seeds = [cases.SEED + 17 * k for k in range(B)]
    T = 9
    t, y, has = cases.poisson_counts(T, missing=0.15)
    b = NativePfBatch(models[0], n, B)
    if whole:
        for k in range(B):
            b.chain(k).set_option(6, whole)        # CSSM_OPT_WHOLE_TILES: the large clouds' geometries at test sizes
    ll, path, rc = b.filter(models, seeds, t, y, has)
    assert not rc.any()
    for k in range(B):
        g = NativePf(models[k], n, seeds[k])
        g.set_option(6, whole)
        gl, _, gess, gpath = g.run(t, y, has, want_path=True)
        assert ll[k] == gl, (k, ll[k], gl)
        np.testing.assert_array_equal(path[k], gpath)
        np.testing.assert_array_equal(b.chain(k).particles(), g.particles())
        np.testing.assert_array_equal(b.chain(k).ancestors(), g.ancestors())
        g.close()
        if n <= 100_000:
            o = oracle.OraclePf(models[k].descriptor(), n, seeds[k])
            ol, _, _, opath = o.filter(t, y, has, want_path=True)
            assert ll[k] == ol
            np.testing.assert_array_equal(path[k], opath)
    # the batch again (buffers, generations and completion words reused), with the chains' seeds rotated
    seeds2 = seeds[1:] + seeds[:1]
    ll2, path2, _ = b.filter(models, seeds2, t, y, has)
    for k in range(B):
        g = NativePf(models[k], n, seeds2[k]); g.set_option(6, whole)
        assert ll2[k] == g.run(t, y, has)[0]
        g.close()
    b.close()


def test_a_chain_with_an_outlying_observation_is_redone_on_its_own():
    """Chain 1's data-independent parameters make observation 4 an outlier for IT only (its level is ruled out by its max): the batch
    holds that chain at it, the others run on; the held chain is then run through the single-handle driver -- every chain's bits are
    still those of a handle of its own."""
    um = cases.c2_unparam()
    B, n, T = 3, 20000, 10
    t, y, has = cases.poisson_counts(T)
    y = y.copy(); y[4] = 60.0
    models = [um.run(_perturbed(cases.c2_params, k)) for k in range(B)]
    seeds = [5 + k for k in range(B)]
    b = NativePfBatch(models[0], n, B)
    ll, path, rc = b.filter(models, seeds, t, y, has)
    for k in range(B):
        o = oracle.OraclePf(models[k].descriptor(), n, seeds[k])
        ol, _, _, opath = o.filter(t, y, has, want_path=True)
        assert rc[k] == 0 and ll[k] == ol
        np.testing.assert_array_equal(path[k], opath)
    b.close()


def test_other_models_and_configurations_fall_back_to_single_handles():
    """LGCP (its first event's level is its max) is not served by the batched launches: the chains run one by one, same bits."""
    model = cases.c4_model()
    B, n = 2, 4000
    t, y, has = cases.event_times(7)
    # (NativePfBatch builds its handles without an LGCP precision: a Poisson-free model needs one -- use the negbin model instead for the
    #  "served" check and a multinomial resampler for the fallback)
    nb = cases.negbin_model()
    tt, yy, hh = cases.poisson_counts(8)
    b = NativePfBatch(nb, n, B)
    ll, path, rc = b.filter([nb, nb], [3, 4], tt, yy, hh)
    for k in range(B):
        o = oracle.OraclePf(nb.descriptor(), n, 3 + k)
        assert ll[k] == o.filter(tt, yy, hh)[0]
    for k in range(B):
        b.chain(k).set_option(2, 2)               # CSSM_OPT_RESAMPLER = multinomial: the batched launches do not serve it
    ll, path, rc = b.filter([nb, nb], [3, 4], tt, yy, hh)
    for k in range(B):
        o = oracle.OraclePf(nb.descriptor(), n, 3 + k, oracle.RESAMPLE_MULTINOMIAL)
        assert ll[k] == o.filter(tt, yy, hh)[0]
    b.close()


def test_batched_pmmh_equals_the_chains_run_one_by_one():
    """cssm_pmmh_run_batched == B x cssm_pmmh_run (and so == the oracle's chain, tests/test_pmmh.py) with the chains' own seeds."""
    from composablestatespacemodels_amd.model import TimedObservation
    from composablestatespacemodels_amd.pmmh import pmmh_native, pmmh_native_batched
    um, init = cases.c2_unparam(), cases.c2_params()
    T, n, iters, B = 30, 6000, 6, 3
    t, y, has = cases.poisson_counts(T, missing=0.1)
    data = [TimedObservation(float(a), float(v) if h else None) for a, v, h in zip(t, y, has)]
    inits = [_perturbed(cases.c2_params, k) for k in range(B)]
    seeds = [20260101 + k for k in range(B)]
    ll, th, acc, last = pmmh_native_batched(um, inits, data, n, 0.01, iters, seeds)
    for k in range(B):
        l1, t1, a1, s1 = pmmh_native(um, inits[k], data, n, 0.01, iters, seed=seeds[k])
        np.testing.assert_array_equal(ll[k], l1); np.testing.assert_array_equal(th[k], t1)
        np.testing.assert_array_equal(acc[k], a1); np.testing.assert_array_equal(last[k], s1)


@pytest.mark.parametrize("iters,delta", [(9, 0.01), (12, 0.25), (1, 0.01)])
def test_speculative_pmmh_is_the_sequential_chain(iters, delta):
    """cssm_pmmh_run_speculative == cssm_pmmh_run: iteration i's proposal and both candidates for iteration i + 1 are filtered as one
    batch of three, the decisions follow -- the chain (log-likelihoods, parameters, acceptance counts, sampled states) bit for bit; an odd
    number of iterations, a single one, and a step size large enough that proposals are rejected wholesale (and some cannot be weighed)."""
    from composablestatespacemodels_amd.model import TimedObservation
    from composablestatespacemodels_amd.pmmh import pmmh_native, pmmh_native_speculative
    um, init = cases.c2_unparam(), cases.c2_params()
    T, n = 30, 6000
    t, y, has = cases.poisson_counts(T, missing=0.1)
    data = [TimedObservation(float(a), float(v) if h else None) for a, v, h in zip(t, y, has)]
    ref = pmmh_native(um, init, data, n, delta, iters, seed=20260105)
    got = pmmh_native_speculative(um, init, data, n, delta, iters, seed=20260105)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    assert 0 < ref[2][-1] or iters == 1 or delta > 0.1      # (the small-step chains accept something: both branches are exercised)

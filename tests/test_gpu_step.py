"""-m gpu: k_step -- the resampling of one observation merged with the propagate of the next into ONE launch
(CSSM_OPT_ONE_LAUNCH = 1; opt-in) -- against the two-launch path (the default) and the oracle: ll, ll_t,
ess_t, the sampled path, the final ancestors, log-weights and clouds must be identical bit for bit.
"""
import numpy as np
import pytest

import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

pytestmark = pytest.mark.gpu

OPT_ONE_LAUNCH = 5


def _run(model, n, t, y, has, one_launch, want_path=True, seed=cases.SEED, exact=0, whole=0):
    g = NativePf(model, n, seed)
    g.set_option(6, whole)                   # CSSM_OPT_WHOLE_TILES
    g.set_option(OPT_ONE_LAUNCH, one_launch)
    if exact:
        g.set_option(1, 1)
    g.profile(True)
    ll, ll_t, ess_t, path = g.run(t, y, has, want_path=want_path)
    prof = g.profile_read()
    out = dict(ll=ll, ll_t=ll_t, ess_t=ess_t, path=path, anc=g.ancestors(), part=g.particles(), logw=g.logw(), prop=g.proposed(),
               merged=prof["k_step"][1], launches=sum(v[1] for v in prof.values()))
    g.close()
    return out


def _same(a, b, what):
    assert a["ll"] == b["ll"], (what, a["ll"], b["ll"])
    for k in ("ll_t", "ess_t", "path", "anc", "part", "prop", "logw"):
        if a[k] is None:
            assert b[k] is None
            continue
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"{what}: {k}")


def _oracle(model, n, t, y, has, seed=cases.SEED):
    o = oracle.OraclePf(model.descriptor(), n, seed)
    ll, ll_t, ess_t, path = o.filter(t, y, has, want_path=True)
    return dict(ll=ll, ll_t=ll_t, ess_t=ess_t, path=path, anc=o.ancestors(), part=o.particles(), logw=o.logw(), prop=o.proposed())


def _pairs(has):
    """weighted observations directly followed by a weighted one: what k_step merges"""
    h = np.asarray(has, dtype=bool)
    return int(np.sum(h[:-1] & h[1:]))


@pytest.mark.parametrize("name,n,T,missing", [
    ("c1_model", 1000, 60, 0.0), ("c1_model", 1, 5, 0.0), ("c1_model", 2, 6, 0.3), ("c2_model", 1023, 9, 0.2), ("c2_model", 1024, 9, 0.0),
    ("c2_model", 1025, 9, 0.2), ("c2_model", 2049, 7, 0.0), ("c2_model", 100_000, 25, 0.1), ("c2_model", 262_145, 6, 0.0),
    ("c3_model", 4096, 20, 0.15), ("linear_model", 30_000, 12, 0.2), ("gen_brownian_seasonal_gaussian", 5000, 12, 0.0),
    ("euler_model", 3000, 10, 0.0), ("negbin_model", 7000, 10, 0.1), ("zip_model", 5000, 8, 0.0), ("max_dim_model", 2049, 8, 0.0),
])
def test_one_launch_equals_two_launches_and_oracle(name, n, T, missing):
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(T, missing=missing)   # (Gaussian models: the counts serve as real-valued observations)
    a = _run(model, n, t, y, has, 1)
    b = _run(model, n, t, y, has, 0)
    assert a["merged"] == _pairs(has) and b["merged"] == 0
    _same(a, b, "one launch vs two")
    o = _oracle(model, n, t, y, has)
    o["logw"] = a["logw"] if not has[-1] else o["logw"]
    _same(a, o, "one launch vs oracle")


@pytest.mark.parametrize("d", list(range(1, 17)))
def test_one_launch_every_dimension(d):
    model = cases.dim_model(d)
    n, T = 3001, 7
    t, y, has = cases.poisson_counts(T, missing=0.15)
    a = _run(model, n, t, y, has, 1)
    assert a["merged"] == _pairs(has)
    o = _oracle(model, n, t, y, has)
    o["logw"] = a["logw"] if not has[-1] else o["logw"]
    _same(a, o, f"d = {d}")


def test_one_launch_range_and_full_bench_size():
    """Not taken unless asked for; -1 = clouds of up to 2^18 particles; 1 = whenever eligible = at most 1024 units of sums: up to
    2^19 particles with the half-tile units of clouds below 2^20, and 2^20 itself (whole units: 1024 of them), never beyond."""
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(8, missing=0.1)
    for n, opt, whole, expect in ((50_000, 0, 0, False), (50_000, -1, 0, True), (1 << 18, -1, 0, True), ((1 << 18) + 1, -1, 0, False),
                                  (1 << 19, 1, 0, True), ((1 << 19) + 1, 1, 0, False), (1 << 20, 1, 0, True), (1 << 20, 1, 2, True),
                                  ((1 << 20) + 1, 1, 2, False)):
        a = _run(model, n, t, y, has, opt, want_path=False, whole=whole)
        assert (a["merged"] > 0) == expect, (n, opt, whole, a["merged"])
        if expect:
            assert a["merged"] == _pairs(has)
            b = _run(model, n, t, y, has, 0, want_path=False)
            _same(a, b, f"N = {n}")
    o = _oracle(model, 1 << 20, t[:4], y[:4], has[:4])
    a = _run(model, 1 << 20, t[:4], y[:4], has[:4], 1, whole=2)
    assert a["merged"] == _pairs(has[:4])
    _same(a, o, "N = 2^20 vs oracle")


def test_one_launch_outlying_observations_hold_and_resume():
    """An observation whose reference level the max rules out (a count of 60 among counts of 0..20) is found by the NEXT
    launch's prologue: the series goes on hold there, the host redoes that observation with the max and the rest is
    enqueued again -- also when the outlier is the first or the last observation."""
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(14)
    y = y.copy(); y[0] = 60.0; y[6] = 75.0; y[7] = 80.0; y[13] = 90.0
    for n in (3000, 1 << 17):
        a = _run(model, n, t, y, has, 1)
        assert a["merged"] >= 9
        _same(a, _oracle(model, n, t, y, has), f"outliers, N = {n}")


def test_one_launch_degenerate_weights_long_runs():
    """A very informative observation puts nearly all weight on a few particles: every block's slots belong to the same few
    parents, and most units have no offspring at all (skipped)."""
    model = cases.linear_model(obs_sd=0.001)
    t = np.array([0.0, 1.0, 2.0, 3.0]); y = np.array([3.0, 3.1, 2.9, 3.05]); has = np.ones(4, dtype=np.uint8)
    for n in (4096, 200_000):
        a = _run(model, n, t, y, has, 1)
        assert a["merged"] == 3
        _same(a, _oracle(model, n, t, y, has), f"degenerate, N = {n}")
        assert a["ess_t"].min() < n // 50


def test_one_launch_forced_exact_offspring_and_reuse_of_a_handle():
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(12, missing=0.1)
    n = 50_000
    g = NativePf(model, n, cases.SEED)
    g.set_option(OPT_ONE_LAUNCH, 1)
    r1 = g.run(t, y, has)
    g.set_option(1, 1)                       # CSSM_OPT_EXACT_OFFSPRING: the contract's exact count for every particle and unit boundary
    r2 = g.run(t, y, has)
    g.set_option(1, 0)
    t2, y2, has2 = cases.poisson_counts(9, seed=5)
    r3 = g.run(t2, y2, has2)                 # the same handle, another series: the two sets of log-weights start wherever they were left
    g.close()
    assert r1[0] == r2[0] and np.array_equal(r1[1], r2[1]) and np.array_equal(r1[2], r2[2])
    assert r1[0] == oracle.OraclePf(model.descriptor(), n, cases.SEED).filter(t, y, has)[0]
    assert r3[0] == oracle.OraclePf(model.descriptor(), n, cases.SEED).filter(t2, y2, has2)[0]


def test_one_launch_then_streaming_continues_the_same_filter():
    """A batch run leaves the handle in the state the two-launch path would: stepping on from it gives the oracle's bits."""
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(10)
    n = 20_000
    g = NativePf(model, n, cases.SEED)
    g.set_option(OPT_ONE_LAUNCH, 1)
    g.run(t[:6], y[:6], has[:6])
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    o.filter(t[:6], y[:6], has[:6])
    for s in range(6, 10):
        assert g.step(t[s], y[s], bool(has[s])) == o.step(t[s], y[s], bool(has[s]))
    np.testing.assert_array_equal(g.particles(), o.particles())
    g.close()


def test_one_launch_halves_the_launches():
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(40)
    a = _run(model, 10_000, t, y, has, 1, want_path=False)
    b = _run(model, 10_000, t, y, has, 0, want_path=False)
    assert b["launches"] == 80 and a["launches"] == 41 and a["merged"] == 39

"""FilterInterpolate (model/ParticleFilter.scala:273-311) and the summary of examples/Interpolate.scala:42-44.

CPU: the oracle's literal restatement (particles ARE paths; a weighted step copies whole paths) against an
independent reconstruction from the per-step ancestor arrays of the plain filter.
GPU: cssm_pf_interpolate (ancestor history on the device, genealogy composed backwards) against the oracle --
order statistics bit-exact, means within 1e-12 (the device sums in a different order)."""
import numpy as np
import pytest

import cases
from oracle import oracle


def _series(name, T):
    if name == "linear_model":
        t, y, has = cases.gaussian_series(T)
    else:
        t, y, has = cases.poisson_counts(T)
    has = has.copy()
    has[T // 3: T // 3 + max(2, T // 5)] = 0          # a block of removed observations, as the example does (:31-33)
    return t, y, has


def _paths_from_ancestors(model, n, t, y, has, seed):
    """Surviving lineages from the plain filter's per-step clouds and ancestors (numpy, independent of the C loop)."""
    o = oracle.OraclePf(model.descriptor(), n, seed)
    o.init(float(np.min(t)))
    clouds, ancs = [o.particles().T.copy()], [np.arange(n)]
    for s in range(len(t)):
        o.step(t[s], y[s], bool(has[s]))
        clouds.append(o.proposed().T.copy())          # x1 of step s
        ancs.append(o.ancestors().astype(np.int64))   # identity on an unweighted step
    b = np.arange(n)
    out = [None] * len(clouds)
    for s in range(len(clouds) - 1, -1, -1):
        b = ancs[s][b]
        out[s] = clouds[s][b]
    return out


@pytest.mark.parametrize("name,n,T", [("c2_model", 500, 20), ("linear_model", 257, 15)])
def test_oracle_path_resampling_equals_ancestor_genealogy(name, n, T):
    model = getattr(cases, name)()
    t, y, has = _series(name, T)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    ll, m, lo, hi, em, el, eu = o.interpolate(t, y, has)
    assert ll == oracle.OraclePf(model.descriptor(), n, cases.SEED).filter(t, y, has)[0]     # same weights as stepFilter
    paths = _paths_from_ancestors(model, n, t, y, has, cases.SEED)
    idx = int(np.floor(0.975 * n))
    for s, cloud in enumerate(paths):
        srt = np.sort(cloud, axis=0)
        np.testing.assert_array_equal(lo[s], srt[n - idx - 1])
        np.testing.assert_array_equal(hi[s], srt[idx - 1])
        np.testing.assert_allclose(m[s], cloud.mean(axis=0), rtol=1e-12, atol=1e-13)
    # the lineages thin out backwards in time: fewer distinct ancestors at the start than at the end
    assert len(np.unique(paths[0][:, 0])) < len(np.unique(paths[-1][:, 0]))
    # where observations were removed the interval is wider than just before the gap
    g0 = T // 3
    width = hi[:, 0] - lo[:, 0]
    assert width[g0 + 2] > 0 and np.all(np.isfinite(width))


def test_reference_pairing_reverses_the_clouds_but_not_the_times():
    model = cases.c2_model()
    T, n = 12, 300
    t, y, has = _series("c2_model", T)
    a = oracle.OraclePf(model.descriptor(), n, cases.SEED).interpolate(t, y, has)
    b = oracle.OraclePf(model.descriptor(), n, cases.SEED).interpolate(t, y, has, reference_pairing=True)
    assert a[0] == b[0]
    for k in (1, 2, 3):                              # state summaries do not depend on the time: exactly reversed
        np.testing.assert_array_equal(a[k], b[k][::-1])
    # eta is evaluated with the time of the OUTPUT index (seasonal f): reversed clouds at another time differ
    assert not np.array_equal(a[5], b[5][::-1])
    assert a[5][T // 2] == b[5][T // 2]              # T even: the middle entry pairs index T/2 with itself


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,T", [("c2_model", 5000, 25), ("c3_model", 2048, 12), ("linear_model", 1 << 16, 10),
                                      ("c1_model", 1000, 40)])
@pytest.mark.parametrize("pairing", [False, True])
def test_gpu_interpolate_matches_oracle(name, n, T, pairing):
    from composablestatespacemodels_amd.filter import NativePf
    model = getattr(cases, name)()
    t, y, has = _series(name, T)
    g = NativePf(model, n, cases.SEED)
    gl, gm, glo, ghi, gem, gel, geu = g.interpolate(t, y, has, 0.975, pairing)
    ol, om, olo, ohi, oem, oel, oeu = oracle.OraclePf(model.descriptor(), n, cases.SEED).interpolate(t, y, has, 0.975, pairing)
    assert gl == ol
    np.testing.assert_array_equal(glo, olo)
    np.testing.assert_array_equal(ghi, ohi)
    np.testing.assert_array_equal(gel, oel)
    np.testing.assert_array_equal(geu, oeu)
    np.testing.assert_allclose(gm, om, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gem, oem, rtol=1e-12, atol=1e-13)
    # the handle is reusable afterwards and the batch filter is unaffected
    assert g.run(t, y, has)[0] == ol
    g.close()


@pytest.mark.gpu
def test_gpu_interpolate_mirror_and_errors():
    from composablestatespacemodels_amd import CssmError, Data
    from composablestatespacemodels_amd.filter import FilterInterpolate, FilterLgcp, NativePf, ParticleFilter, Resampling
    model = cases.c2_model()
    t, y, has = _series("c2_model", 30)
    data = [Data(float(a), float(b) if h else None) for a, b, h in zip(t, y, has)]
    ll, out = FilterInterpolate(model, Resampling.systematicResampling).interpolate(data, 4096)
    assert len(out) == 31 and out[0].observation is None and out[0].time == 0.0 and out[11].observation is None
    assert all(o.etaIntervals.lower <= o.etaIntervals.upper for o in out)
    assert all(ci.lower < m < ci.upper for o in out for ci, m in zip(o.stateIntervals, o.state))
    ll2, out2 = ParticleFilter.interpolate(Resampling.systematicResampling, 0.0, 4096)(model)(data)
    assert ll2 == ll and out2[7].eta == out[7].eta
    g = NativePf(model, 100, 1)
    with pytest.raises(CssmError):
        g.interpolate(t, y, has, 1.5)
    with pytest.raises(CssmError):
        g.step(1.0, 1.0)                              # streaming needs cssm_pf_init first
    g.close()
    tt, yy, hh = cases.event_times(5)
    l = NativePf(cases.c4_model(), 100, 1, lgcp_precision=1)
    with pytest.raises(CssmError):
        l.interpolate(tt, yy, hh)
    l.close()


@pytest.mark.gpu
def test_gpu_interpolate_with_outlying_observation():
    from composablestatespacemodels_amd.filter import NativePf
    model = cases.c2_model()
    n, T = 3000, 12
    t, y, has = _series("c2_model", T)
    y = y.copy(); y[1] = 60.0
    want = oracle.OraclePf(model.descriptor(), n, cases.SEED).interpolate(t, y, has)
    for fused in (0, 1):
        g = NativePf(model, n, cases.SEED)
        g.set_option(3, fused)
        got = g.interpolate(t, y, has)
        assert got[0] == want[0]
        for k in (2, 3, 5, 6):
            np.testing.assert_array_equal(got[k], want[k])
        g.close()

"""-m gpu: k_offspring_wave (round 6: the resampling kernel that takes its waves' prefixes from the exact sums the propagate's waves left and
runs the rest in fp64, CSSM_OPT_WAVE_SUMS) against k_offspring_self (every weight converted and scanned in 128 bits) and the oracle: the same
bits -- ll after every observation, ESS, ancestors, clouds -- on whole and ragged clouds, one and several chunks per wave, both group-sum
layouts, with every chunk forced through the exact path, across outlying observations (redone by the old kernels in place) and
continued calls."""
import numpy as np
import pytest

import cases
from composablestatespacemodels_amd.filter import NativePf
from oracle import oracle

pytestmark = pytest.mark.gpu
OPT_EXACT, OPT_WHOLE_TILES, OPT_WAVE_SUMS = 1, 6, 10


def _run(model, n, t, y, has, wave, exact=0, whole=0, cut=None):
    g = NativePf(model, n, cases.SEED)
    g.set_option(OPT_WAVE_SUMS, 2 if wave else 0); g.set_option(OPT_EXACT, exact)   # (2: the mapping wherever the geometry allows, also at one tile per unit)
    if whole:
        g.set_option(OPT_WHOLE_TILES, whole)
    if cut is None:
        ll, ll_t, ess, _ = g.run(t, y, has)
    else:
        _, a, b, _ = g.run(t[:cut], y[:cut], has[:cut])
        ll, c, d = g.run_more(t[cut:], y[cut:], has[cut:])
        ll_t, ess = np.concatenate([a, c]), np.concatenate([b, d])
    out = (ll, ll_t, ess, g.ancestors(), g.particles())
    g.close()
    return out


def _same(a, b, what):
    assert a[0] == b[0], what
    np.testing.assert_array_equal(a[1], b[1], err_msg=what)
    np.testing.assert_array_equal(a[2], b[2], err_msg=what)
    np.testing.assert_array_equal(a[3], b[3], err_msg=what)
    np.testing.assert_array_equal(a[4], b[4], err_msg=what)


@pytest.mark.parametrize("name,n,whole", [("c2_model", 64 * 1024, 1), ("c2_model", 65 * 1024 + 3, 1), ("c1_model", 200 * 1024 + 511, 1),
                                          ("c3_model", 100 * 1024 + 1, 1), ("c2_model", 1 << 20, 0), ("c1_model", (1 << 20) + 1, 0)])
def test_wave_sums_kernel_equals_the_128_bit_kernel_and_the_oracle(name, n, whole):
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(9, missing=0.2)
    y = y.copy(); y[5] = 70.0; has[5] = 1                      # an outlying observation: held, redone relative to the max, the series goes on
    new = _run(model, n, t, y, has, 1, whole=whole, cut=4)
    old = _run(model, n, t, y, has, 0, whole=whole, cut=4)
    _same(new, old, "wave sums on / off")
    for mode in (1, 2):                                       # every chunk through the contract's exact predicate
        _same(_run(model, n, t, y, has, 1, exact=mode, whole=whole), new, f"exact mode {mode}")
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    ol, oll, oess, _ = o.filter(t, y, has)
    assert new[0] == ol
    np.testing.assert_array_equal(new[1], oll)
    np.testing.assert_array_equal(new[2], oess)
    np.testing.assert_array_equal(new[3], o.ancestors())
    np.testing.assert_array_equal(new[4], o.particles())


@pytest.mark.parametrize("name,n", [("c2_model", (3 << 20) + 777), ("c1_model", (1 << 22) + 12345), ("c1_model", 1 << 23), ("c2_model", (1 << 24) - 4097)])
def test_wave_sums_kernel_equals_the_128_bit_kernel_at_large_sizes(name, n):
    """Several chunks per wave (units of 2 .. 4 tiles), ragged last units, layout 2 of the group sums from 2^22 + 1 particles on."""
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(6, missing=0.2)
    new = _run(model, n, t, y, has, 1, cut=3)
    _same(new, _run(model, n, t, y, has, 0, cut=3), "wave sums on / off")
    _same(_run(model, n, t, y, has, 1, exact=1), new, "every chunk exact")
    assert np.all(np.diff(new[3].astype(np.int64)) >= 0)


@pytest.mark.parametrize("yout", [55.0, 60.0, 64.0])
def test_wave_sums_kernel_on_a_level_far_above_the_max(yout):
    """An observation whose reference level sits 20-32 above the largest log-weight is ACCEPTED (cssm_ref_choose): every weight is tiny, S_tot a
    few powers of two above its floor -- the case the fp64 prefixes' error band widens for (eps = N (2^-44 + 2 q 2^-96 / S_tot))."""
    model = cases.c2_model()
    n = 70 * 1024 + 5
    t, y, has = cases.poisson_counts(8)
    y = y.copy(); y[4] = yout; has[4] = 1
    new = _run(model, n, t, y, has, 1, whole=1)
    _same(new, _run(model, n, t, y, has, 0, whole=1), "wave sums on / off")
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    ol, oll, oess, _ = o.filter(t, y, has)
    assert new[0] == ol
    np.testing.assert_array_equal(new[2], oess)
    np.testing.assert_array_equal(new[3], o.ancestors())


def _two_component_model_outside_the_table():
    """poisson(ouProcess(2)): d = 2, a structure the library holds no ahead-of-time kernel for -- its propagate is compiled at run time."""
    from composablestatespacemodels_amd import Model, Parameters, Sde, SdeParameter
    return Model.poisson(Sde.ouProcess(2)).run(Parameters.apply(None, SdeParameter.ouParameter(0.0, 1.0, 0.2, [0.5, -0.25], 0.3)))


def test_wave_ranges_in_a_runtime_compiled_propagate():
    """The wave-range mapping is a flag of the LAUNCH, so a propagate kernel compiled at run time (hipRTC: a model outside the table) takes it
    like the ahead-of-time ones: the DEFAULT path of a d = 2 cloud beyond 2^20 particles (k_offspring_wave behind it) against the 128-bit
    kernel bit for bit, and -- at a size the one-thread oracle affords, the mapping forced -- against the oracle."""
    from composablestatespacemodels_amd import _abi
    import ctypes as C
    model = _two_component_model_outside_the_table()
    t, y, has = cases.poisson_counts(6, missing=0.2)
    info = (C.c_uint64 * 4)()       # compiled, from disk, launches, failures
    _abi.check(_abi.load_library().cssm_rtc_info(info)); before = list(info)
    n = (1 << 21) + 4099

    def run(ws):
        g = NativePf(model, n, cases.SEED)
        if ws is not None:
            g.set_option(OPT_WAVE_SUMS, ws)
        ll, ll_t, ess, _ = g.run(t, y, has)
        out = (ll, ll_t, ess, g.ancestors(), g.particles()); g.close()
        return out
    default = run(None)
    _abi.check(_abi.load_library().cssm_rtc_info(info))
    assert info[2] - before[2] == len(t) and info[3] == before[3] == 0, "the model was expected to run on runtime-compiled kernels"
    _same(default, run(0), "default path (wave ranges) / 128-bit kernel")
    _same(default, run(2), "default path / wave ranges forced")
    n2 = 70 * 1024 + 3
    new = _run(model, n2, t, y, has, 1, whole=1)
    o = oracle.OraclePf(model.descriptor(), n2, cases.SEED)
    ol, oll, oess, _ = o.filter(t, y, has)
    assert new[0] == ol
    np.testing.assert_array_equal(new[2], oess)
    np.testing.assert_array_equal(new[3], o.ancestors())
    np.testing.assert_array_equal(new[4], o.particles())


def test_streaming_steps_on_the_default_path_beyond_2_20_particles():
    """cssm_pf_step (one observation per call, the shape of the reference's Flow.scan) at d = 1 beyond 2^20 particles takes the wave-range
    propagate + k_offspring_wave by default: the same bits as with the mapping switched off, observation by observation."""
    model = cases.c1_model()
    n = (1 << 21) + 17
    t, y, has = cases.poisson_counts(5, missing=0.2)

    def run(ws):
        g = NativePf(model, n, cases.SEED)
        if ws is not None:
            g.set_option(OPT_WAVE_SUMS, ws)
        g.init(float(np.min(t)))
        out = []
        for s in range(len(t)):
            ll, ess = g.step(float(t[s]), float(y[s]), bool(has[s]))
            out.append((ll, ess, g.ancestors().copy()))
        cloud = g.particles(); g.close()
        return out, cloud
    a, ca = run(None)
    b, cb = run(0)
    for s, (x, z) in enumerate(zip(a, b)):
        assert x[0] == z[0] and x[1] == z[1], s
        np.testing.assert_array_equal(x[2], z[2])
    np.testing.assert_array_equal(ca, cb)

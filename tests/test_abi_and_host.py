"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/cssm_pf.h
declares, fails loudly without a GPU (no CPU fallback), and the host-side mirror of the reference's
model interface builds the descriptors the reference's constructors imply."""
import ctypes as C
import math
import os
import re
import subprocess

import numpy as np
import pytest

import cases
from composablestatespacemodels_amd import (CssmError, Model, Parameters, Sde, SdeParameter, _abi, load_library,
                                            logistic)
from composablestatespacemodels_amd.filter import Filter, ParticleFilter, Resampling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "cssm_pf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cssm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libcssm_pf.so does not export {n}"
    bound = {s[0] for s in _abi.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound
    assert b"gfx950" in lib.cssm_version()


def test_library_contains_gfx950_code_object_and_no_oracle():
    out = subprocess.run(["strings", "-n", "6", _abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdgcn-amd-amdhsa--gfx950" in out
    assert "oracle_pf_" not in out          # the product never links the test oracle


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for hosts without a GPU")
def test_no_cpu_fallback_fails_loudly_without_gpu():
    lib = load_library()
    h = C.c_void_p()
    rc = lib.cssm_pf_create(cases.c1_model().descriptor().ptr(), 100, 1, 0, C.byref(h))
    assert rc == _abi.CSSM_EHIP and not h.value
    assert b"no CPU path" in lib.cssm_last_error()
    w = np.ones(8); anc = np.zeros(8, dtype=np.uint32)
    rc = lib.cssm_resample_systematic(w.ctypes.data_as(C.POINTER(C.c_double)), 8, 0.5,
                                      anc.ctypes.data_as(C.POINTER(C.c_uint32)), 0)
    assert rc == _abi.CSSM_EHIP
    from composablestatespacemodels_amd import Data
    with pytest.raises(CssmError):
        Filter(cases.c1_model(), Resampling.systematicResampling).llFilter([Data(0.0, 1.0), Data(1.0, 2.0)], 10)


def test_missing_library_is_an_import_error(tmp_path):
    with pytest.raises(ImportError):
        load_library(str(tmp_path / "libcssm_pf.so"))


def test_desc_flatten_matches_reference_order():
    lib = load_library()
    m = cases.gen_brownian_seasonal_gaussian()
    d = m.descriptor()
    n = C.c_size_t()
    assert lib.cssm_desc_flatten(d.ptr(), None, 0, C.byref(n)) == 0
    theta = np.zeros(n.value)
    assert lib.cssm_desc_flatten(d.ptr(), theta.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)) == 0
    # Parameters.flattenParams (Parameters.scala:88-95): scale first, then m0 ++ c0 ++ [phi] ++ mu ++ sigma per leaf
    np.testing.assert_allclose(theta, m.parameters().flattenParams(), rtol=0, atol=0)
    assert theta[0] == math.log(0.7) and theta[1] == 0.2 and theta[2] == math.log(1.5)


def test_invalid_descriptors_are_rejected_before_touching_the_gpu():
    # the reference returns Failure(...) for mismatched parameters (Sde.scala:183,188,201)
    with pytest.raises(ValueError):
        Model.poisson(Sde.ouProcess(1)).run(Parameters.apply(None, SdeParameter.brownianParameter(0, 1, 1)))
    with pytest.raises(ValueError):   # seasonal needs 2*harmonics components (Model.scala:217-225)
        Model.seasonal(24, 2, Sde.ouProcess(3)).run(Parameters.apply(1.0, SdeParameter.ouParameter(0, 1, .2, 0, 1)))
    with pytest.raises(ValueError):   # "Must provide SD parameter" (Model.scala:250)
        Model.linear(Sde.brownianMotion(1)).run(Parameters.apply(None, SdeParameter.brownianParameter(0, 1, 1)))
    with pytest.raises(ValueError):   # branch parameters for a leaf model (Model.scala:47)
        Model.poisson(Sde.brownianMotion(1)).run(cases.c2_params())


def test_smart_constructors_store_unconstrained_values():
    p = SdeParameter.ouParameter(0.5, 2.0, 0.2, 1.0, 0.3)       # SdeParameters.scala:202-205
    assert p.c0 == [math.log(2.0)] and p.sigma == [math.log(0.3)] and p.phi == [logistic(0.2)] and p.mu == [1.0]
    q = SdeParameter.brownianParameter([0.0, 1.0], 1.0, 0.01)   # :197-200
    assert q.flatten() == [0.0, 1.0, 0.0, math.log(0.01)]
    r = SdeParameter.ouParameterUnconstrained(0.5, 2.0, 0.2, 1.0, 0.3)
    assert r.flatten() == [0.5, 2.0, 0.2, 1.0, 0.3]


def test_composition_is_left_nested_and_leftmost_model_observes():
    m = cases.c3_model()
    assert m.dimension == 9 and len(m.leaves) == 2 and m.obs_kind == _abi.OBS_POISSON
    d = m.descriptor().desc
    assert d.n_leaves == 2 and d.leaves[0].sde_kind == _abi.SDE_BROWNIAN and d.leaves[1].f_kind == _abi.F_SEASONAL
    assert d.leaves[1].period == 24 and d.leaves[1].harmonics == 4 and d.leaves[1].dim == 8
    g = cases.gen_brownian_seasonal_gaussian()
    assert g.obs_kind == _abi.OBS_GAUSSIAN and g.descriptor().desc.leaves[0].has_scale == 1
    theta = g.parameters().flattenParams()
    assert g.parameters().withFlat(theta).flattenParams() == theta
    assert g.parameters().add([1.0] * len(theta)).flattenParams() == [v + 1.0 for v in theta]


def test_filter_accepts_any_resample_function():
    """`Resample[A]` is a function type (model/package.scala:23): the three native resamplers keep the step on the device, any
    other function is applied on the host (cssm_pf_propagate / cssm_pf_adopt) -- among them residualResampling, which the reference
    cannot run as written (model/Resampling.scala:144-145) and this package offers in its documented intent, as a labelled extension."""
    f = Filter(cases.c1_model(), lambda p, w: p)
    assert f._host_resample is not None and Filter(cases.c1_model(), Resampling.systematicResampling)._host_resample is None
    assert Filter(cases.c1_model(), Resampling.residualResampling)._host_resample is Resampling.residualResampling
    assert "EXTENSION" in Resampling.residualResampling.__doc__
    with pytest.raises(TypeError):
        Filter(cases.c1_model(), 3)
    assert ParticleFilter.effectiveSampleSize([1.0, 1.0, 1.0, 1.0]) == 4      # ParticleFilter.scala:431-434
    assert ParticleFilter.effectiveSampleSize([1.0, 0.0, 0.0, 0.0]) == 1
    assert ParticleFilter.mean([1.0, 2.0, 3.0]) == 2.0


def test_known_model_structures_are_the_words_the_kernels_hold_at_compile_time():
    """csrc/cssm_prop.hip launches k_propagate instantiations that hold a model's structure words at compile time when the handle's
    words ARE those constants (KnownStructures; the LGCP launch's 0x36).  If the packing of ModelK::comp ever changes, the table
    would silently stop matching -- a performance loss no parity test sees: the reference's example models must still map to it."""
    import re
    lib = _abi.load_library()
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "composablestatespacemodels_amd", "csrc", "cssm_prop.hip")).read()
    table = {}
    for dm, body in re.findall(r"template <> struct KnownStructures<(\d+)> \{[^\n]*w\[\d+\]\[3\] = \{(.*)\}; \};", src):
        table[int(dm)] = [tuple(int(x.rstrip("u"), 16) for x in grp.split(",")) for grp in re.findall(r"\{([^{}]*)\}", body)]
    assert set(table) == {1, 3, 9}, table

    def words(model, prec=0):
        w = (C.c_uint32 * 4)()
        d = C.c_int32()
        assert lib.cssm_model_structure(model.descriptor(prec).ptr(), w, C.byref(d)) == 0
        return d.value, tuple(w)[:3]

    for model, prec in ((cases.c1_model(), 0), (cases.c2_model(), 0), (cases.c3_model(), 0), (cases.linear_model(), 0)):
        d, w = words(model, prec)
        assert w in table[d], (d, [hex(x) for x in w])
    d, w = words(cases.c4_model(), 2)            # configs[3]: the LGCP launch compares comp[0] with 0x36
    assert (d, w[0]) == (1, 0x36) and "a.mk.comp[0] == 0x36u" in src
    d, w = words(cases.dim_model(2))              # (a structure outside the table keeps the generic kernels)
    assert w not in table.get(d, [])

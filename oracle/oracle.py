"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement of the reference filter.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; nothing under composablestatespacemodels_amd/ may import this module.
See the header of cssm_oracle.c for the parity status ("parity unpinned by the reference").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CSSM_ORACLE_LIB: another build of the same source (the sanitizer job loads oracle/liboracle_san.so, tests/test_sanitizers.py)
LIB = os.environ.get("CSSM_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")

LITERAL_SUMS = 1
TIE_LAST = 2
LIBM = 4
RESAMPLE_STRATIFIED = 8
RESAMPLE_MULTINOMIAL = 16

OK, EINVAL, ENONFINITE, EEMPTY = 0, -1, -5, -8

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "cssm_oracle.c")
    hdrs = [os.path.join(HERE, "..", "include", h) for h in ("cssm_numerics.h", "cssm_pf.h")]
    if os.environ.get("CSSM_ORACLE_LIB"):
        return LIB
    stale = force or not os.path.exists(LIB) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB) for s in [src] + hdrs)
    if stale and os.path.exists(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        sig = {
            "oracle_pf_create": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]),
            "oracle_pf_destroy": (None, [vp]),
            "oracle_pf_set_params": (C.c_int, [vp, vp]),
            "oracle_pf_reseed": (None, [vp, C.c_uint64]),
            "oracle_pf_components": (C.c_int, [vp, _dp]),
            "oracle_pf_dump_normals": (None, [vp, C.c_uint32, C.c_int, C.c_int, _dp]),
            "oracle_pf_init": (C.c_int, [vp, C.c_double]),
            "oracle_pf_init_from": (C.c_int, [vp, C.c_double, _dp]),
            "oracle_pf_step": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, _dp, _i32p]),
            "oracle_pf_filter": (C.c_int, [vp, _dp, _dp, _u8p, C.c_size_t, _dp, _dp, _i32p, _dp]),
            "oracle_pf_set_shard": (None, [vp, C.c_uint64, C.c_uint64]),
            "oracle_pf_propagate_only": (C.c_int, [vp, C.c_double, C.c_double, C.c_int]),
            "oracle_pf_set_particles": (None, [vp, _dp]),
            "oracle_pf_num_particles": (C.c_uint64, [vp]),
            "oracle_pf_dim": (C.c_int, [vp]),
            "oracle_pf_get_particles": (None, [vp, _dp]),
            "oracle_pf_get_proposed": (None, [vp, _dp]),
            "oracle_pf_get_logw": (None, [vp, _dp]),
            "oracle_pf_eta": (None, [vp, _dp]),
            "oracle_pf_eta_of": (C.c_double, [vp, _dp]),
            "oracle_pf_get_ancestors": (None, [vp, _u32p]),
            "oracle_pf_get_cumw": (None, [vp, _dp]),
            "oracle_pf_get_ref": (None, [vp, _dp, _dp]),
            "oracle_pf_ref_level": (C.c_double, [vp, C.c_double]),
            "oracle_c_ref_choose": (C.c_double, [C.c_double, C.c_double]),
            "oracle_c_ref_predict": (C.c_double, [C.c_double]),
            "oracle_c_strat_count": (C.c_uint64, [C.c_double, C.c_uint64, C.c_uint32, C.c_uint64]),
            "oracle_c_order_key": (C.c_uint64, [C.c_double]),
            "oracle_c_order_unkey": (C.c_double, [C.c_uint64]),
            "oracle_pf_summary": (C.c_int, [vp, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp]),
            "oracle_pf_interpolate": (C.c_int, [vp, _dp, _dp, _u8p, C.c_size_t, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
            "oracle_resample_systematic": (C.c_int, [_dp, C.c_uint64, C.c_double, _u32p, _dp, C.c_int]),
            "oracle_resample_stratified": (C.c_int, [_dp, C.c_uint64, C.c_uint64, C.c_uint32, _u32p, _dp]),
            "oracle_resample_multinomial": (C.c_int, [_dp, C.c_uint64, C.c_uint64, C.c_uint32, _u32p, _dp]),
            "oracle_resample_residual": (C.c_int, [_dp, C.c_uint64, C.c_uint64, C.c_uint32, _u32p]),
            "oracle_logdens_poisson": (C.c_double, [C.c_double, C.c_double]),
            "oracle_logdens_gaussian": (C.c_double, [C.c_double, C.c_double, C.c_double]),
            "oracle_logdens_negbin": (C.c_double, [C.c_double, C.c_double, C.c_double]),
            "oracle_logdens_zip": (C.c_double, [C.c_double, C.c_double, C.c_double]),
            "oracle_logdens_bernoulli": (C.c_double, [C.c_double, C.c_double]),
            "oracle_logdens_student_t": (C.c_double, [C.c_double, C.c_double, C.c_double, C.c_int]),
            "oracle_logdens_beta": (C.c_double, [C.c_double, C.c_double]),
            "oracle_c_lgamma": (C.c_double, [C.c_double]),
            "oracle_pmmh_run": (C.c_int, [vp, vp, _dp, C.c_size_t, C.c_double, _dp, _dp, _u8p, C.c_size_t,
                                          C.c_uint64, C.c_size_t, _dp, _dp, _i32p, _dp]),
            "oracle_c_exp": (C.c_double, [C.c_double]),
            "oracle_c_log": (C.c_double, [C.c_double]),
            "oracle_c_philox": (None, [_u32p, _u32p, _u32p]),
            "oracle_c_philox_contract": (None, [_u32p, _u32p, _u32p]),
            "oracle_c_pick": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint64]),
            "oracle_c_normals": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp]),
            "oracle_c_log_unit_v": (None, [_dp, _dp, C.c_size_t]),
            "oracle_c_fix_v": (None, [_dp, C.POINTER(C.c_uint64), C.c_size_t]),
            "oracle_c_paired_normals_v": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _dp, C.c_size_t]),
            "oracle_c_lgamma_kp1": (C.c_double, [C.c_longlong]),
            "oracle_c_sys_count": (C.c_uint64, [C.c_double, C.c_double, C.c_uint64]),
            "oracle_c_fix_roundtrip": (C.c_double, [C.c_double]),
            "oracle_c_u": (C.c_double, [C.c_uint64, C.c_uint32]),
            "oracle_c_mh_u": (C.c_double, [C.c_uint64, C.c_uint64]),
            "oracle_c_derive_key": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "oracle_c_mh_normals": (None, [C.c_uint64, C.c_uint64, C.c_uint64, _dp]),
            "oracle_c_strat_u_v": (None, [C.c_uint64, C.c_uint32, C.c_uint64, _dp]),
            "oracle_c_multi_u_v": (None, [C.c_uint64, C.c_uint32, C.c_uint64, _dp]),
            "oracle_c_exp_v": (None, [_dp, _dp, C.c_size_t]),
            "oracle_c_log_v": (None, [_dp, _dp, C.c_size_t]),
            "oracle_c_sincos2pi_v": (None, [_dp, _dp, _dp, C.c_size_t]),
            "oracle_c_sincos_u24_v": (None, [C.POINTER(C.c_uint32), _dp, _dp, C.c_size_t]),
            "oracle_c_normals_v": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp, C.c_size_t]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ty=_dp):
    return a.ctypes.data_as(ty)


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


class OraclePf:
    """Sequential CPU filter; the same method names as the native handle wrapper."""

    def __init__(self, descriptor, n: int, seed: int, flags: int = 0):
        self._d = descriptor  # keeps the ctypes arrays alive
        self._h = C.c_void_p()
        _chk(lib().oracle_pf_create(C.cast(descriptor.ptr(), C.c_void_p), n, seed, flags, C.byref(self._h)))
        self.n = n
        self.d = lib().oracle_pf_dim(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().oracle_pf_destroy(self._h)
            self._h = C.c_void_p()

    def set_params(self, descriptor):
        self._d = descriptor
        _chk(lib().oracle_pf_set_params(self._h, C.cast(descriptor.ptr(), C.c_void_p)))

    def reseed(self, seed):
        lib().oracle_pf_reseed(self._h, seed)

    def components(self):
        out = np.zeros((self.d, 5))
        lib().oracle_pf_components(self._h, _p(out))
        return out

    def dump_normals(self, step, init=False, sub=-1):
        """The d normals of every particle at (step, initial draw / ordinary step, LGCP sub-step or -1) in the handle's mode: [n][d]."""
        out = np.zeros((self.n, self.d))
        lib().oracle_pf_dump_normals(self._h, int(step), 1 if init else 0, int(sub), _p(out))
        return out

    def init(self, t0):
        _chk(lib().oracle_pf_init(self._h, t0))

    def init_from(self, t0, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        _chk(lib().oracle_pf_init_from(self._h, t0, _p(s)))

    def step(self, t, y, has_obs=True):
        ll, ess = C.c_double(), C.c_int32()
        _chk(lib().oracle_pf_step(self._h, t, 0.0 if y is None else y, 1 if has_obs else 0, C.byref(ll), C.byref(ess)))
        return ll.value, ess.value

    def filter(self, t, y, has=None, want_path=False):
        t = np.ascontiguousarray(t, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hp = None
        if has is not None:
            has = np.ascontiguousarray(has, dtype=np.uint8); hp = _p(has, _u8p)
        ll = C.c_double(); ll_t = np.zeros(T); ess_t = np.zeros(T, dtype=np.int32)
        path = np.zeros((T + 1, self.d)) if want_path else None
        _chk(lib().oracle_pf_filter(self._h, _p(t), _p(y), hp, T, C.byref(ll), _p(ll_t), _p(ess_t, _i32p),
                                    _p(path) if want_path else None))
        return ll.value, ll_t, ess_t, path

    def set_shard(self, first, n_global):
        lib().oracle_pf_set_shard(self._h, first, n_global)

    def propagate_only(self, t, y, has_obs=True):
        _chk(lib().oracle_pf_propagate_only(self._h, t, 0.0 if y is None else y, 1 if has_obs else 0))

    def set_particles(self, soa):
        a = np.ascontiguousarray(soa, dtype=np.float64)
        assert a.shape == (self.d, self.n)
        lib().oracle_pf_set_particles(self._h, _p(a))

    def summary(self, interval=0.975):
        m, lo, hi = np.zeros(self.d), np.zeros(self.d), np.zeros(self.d)
        em, el, eu = C.c_double(), C.c_double(), C.c_double()
        _chk(lib().oracle_pf_summary(self._h, interval, _p(m), _p(lo), _p(hi), C.byref(em), C.byref(el), C.byref(eu)))
        return m, lo, hi, em.value, el.value, eu.value

    def eta(self):
        out = np.zeros(self.n); lib().oracle_pf_eta(self._h, _p(out)); return out

    def eta_of(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        return float(lib().oracle_pf_eta_of(self._h, _p(s)))

    def interpolate(self, t, y, has=None, interval=0.975, reference_pairing=False):
        t = np.ascontiguousarray(t, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hasa = np.ones(T, dtype=np.uint8) if has is None else np.ascontiguousarray(has, dtype=np.uint8)
        m, lo, hi = (np.zeros((T + 1, self.d)) for _ in range(3))
        em, el, eu = (np.zeros(T + 1) for _ in range(3))
        ll = np.zeros(1)
        _chk(lib().oracle_pf_interpolate(self._h, _p(t), _p(y), _p(hasa, _u8p), T, interval, int(reference_pairing), _p(ll),
                                         _p(m), _p(lo), _p(hi), _p(em), _p(el), _p(eu)))
        return float(ll[0]), m, lo, hi, em, el, eu

    def particles(self):
        out = np.zeros((self.d, self.n)); lib().oracle_pf_get_particles(self._h, _p(out)); return out

    def proposed(self):
        out = np.zeros((self.d, self.n)); lib().oracle_pf_get_proposed(self._h, _p(out)); return out

    def logw(self):
        out = np.zeros(self.n); lib().oracle_pf_get_logw(self._h, _p(out)); return out

    def ancestors(self):
        out = np.zeros(self.n, dtype=np.uint32); lib().oracle_pf_get_ancestors(self._h, _p(out, _u32p)); return out

    def ref_level(self, y):
        return float(lib().oracle_pf_ref_level(self._h, float(y)))

    def ref(self):
        """(rescaling level, max log-weight) of the last weighted step."""
        r, g = np.zeros(1), np.zeros(1)
        lib().oracle_pf_get_ref(self._h, _p(r), _p(g))
        return float(r[0]), float(g[0])

    def cumw(self):
        out = np.zeros(self.n); lib().oracle_pf_get_cumw(self._h, _p(out)); return out

    def pmmh(self, descriptor, theta0, delta, t, y, has, seed, n_iters):
        """descriptor's parameter arrays are overwritten (it is the proposal scratch)."""
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        T, nt = len(t), len(theta0)
        hp = None
        if has is not None:
            has = np.ascontiguousarray(has, dtype=np.uint8); hp = _p(has, _u8p)
        ll = np.zeros(n_iters); th = np.zeros((n_iters, nt)); acc = np.zeros(n_iters, dtype=np.int32)
        last = np.zeros((n_iters, self.d))
        self._d = descriptor
        _chk(lib().oracle_pmmh_run(self._h, C.cast(descriptor.ptr(), C.c_void_p), _p(theta0), nt, delta, _p(t), _p(y),
                                   hp, T, seed, n_iters, _p(ll), _p(th), _p(acc, _i32p), _p(last)))
        return ll, th, acc, last


def seam_scale(w):
    """What the stateless `Resample[A]` seam does with weights of ANY scale before they enter the fixed-point sums (2^-96 grid): the
    reference normalises w / sum(w) (Resampling.scala:21-24), so only ratios matter -- a vector whose largest weight is below 2^-32 is
    multiplied by the power of two that brings that weight into [0.5, 1) (exact).  Mirrors cssm_resample (csrc/cssm_pf.hip)."""
    w = np.ascontiguousarray(w, dtype=np.float64)
    m = float(np.max(w)) if len(w) else 0.0
    if 0.0 < m < 2.0 ** -32:
        return np.ldexp(w, -int(np.frexp(m)[1]))
    return w


def resample_systematic(w, u, flags=0, want_cumw=False):
    w = seam_scale(w) if not (flags & LITERAL_SUMS) else np.ascontiguousarray(w, dtype=np.float64)
    anc = np.zeros(len(w), dtype=np.uint32)
    Cw = np.zeros(len(w))
    rc = lib().oracle_resample_systematic(_p(w), len(w), u, _p(anc, _u32p), _p(Cw), flags)
    _chk(rc)
    return (anc, Cw) if want_cumw else anc


def resample_stratified(w, seed, step=0):
    w = seam_scale(w); anc = np.zeros(len(w), dtype=np.uint32)
    _chk(lib().oracle_resample_stratified(_p(w), len(w), seed, step, _p(anc, _u32p), None)); return anc


def resample_multinomial(w, seed, step=0):
    w = seam_scale(w); anc = np.zeros(len(w), dtype=np.uint32)
    _chk(lib().oracle_resample_multinomial(_p(w), len(w), seed, step, _p(anc, _u32p), None)); return anc


def resample_residual(w, seed, step=0):
    """EXTENSION: residual resampling in its documented intent (oracle_resample_residual; twin of cssm_resample_residual)."""
    w = seam_scale(w); anc = np.zeros(len(w), dtype=np.uint32)
    _chk(lib().oracle_resample_residual(_p(w), len(w), seed, step, _p(anc, _u32p))); return anc


def c_exp(x):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.empty_like(x); lib().oracle_c_exp_v(_p(x), _p(y), x.size); return y


def c_log(x):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.empty_like(x); lib().oracle_c_log_v(_p(x), _p(y), x.size); return y


def c_sincos2pi(u):
    u = np.ascontiguousarray(u, dtype=np.float64); s = np.empty_like(u); c = np.empty_like(u)
    lib().oracle_c_sincos2pi_v(_p(u), _p(s), _p(c), u.size); return s, c


def c_sincos_u24(k):
    k = np.ascontiguousarray(k, dtype=np.uint32); s = np.empty(k.size); c = np.empty(k.size)
    lib().oracle_c_sincos_u24_v(k.ctypes.data_as(C.POINTER(C.c_uint32)), _p(s), _p(c), k.size); return s, c


def c_philox(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32); key = np.asarray(key, dtype=np.uint32); out = np.zeros(4, dtype=np.uint32)
    lib().oracle_c_philox(_p(ctr, _u32p), _p(key, _u32p), _p(out, _u32p)); return out


def c_philox_contract(ctr, key):
    """The generator every variate of the contract comes from (Philox4x32 with CSSM_PHILOX_ROUNDS = 7 rounds)."""
    ctr = np.asarray(ctr, dtype=np.uint32); key = np.asarray(key, dtype=np.uint32); out = np.zeros(4, dtype=np.uint32)
    lib().oracle_c_philox_contract(_p(ctr, _u32p), _p(key, _u32p), _p(out, _u32p)); return out


def c_normals(seed, gid0, step, tag, pair, n):
    z = np.zeros((n, 2)); lib().oracle_c_normals_v(seed, gid0, step, tag, pair, _p(z), n); return z


def c_fix(w):
    """cssm_fix_from_double: (lo, hi) 64-bit words per input."""
    w = np.ascontiguousarray(w, dtype=np.float64); out = np.zeros((w.size, 2), dtype=np.uint64)
    lib().oracle_c_fix_v(_p(w), _p(out, C.POINTER(C.c_uint64)), w.size); return out


def c_paired_normals(seed, gid0, step, tag, d, n):
    """The d normals of each of the particles gid0 .. gid0+n-1 of an ordinary step / the initial draw (pair streams)."""
    z = np.zeros((n, d)); lib().oracle_c_paired_normals_v(seed, gid0, step, tag, d, _p(z), n); return z


def c_log_unit(x):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.empty_like(x); lib().oracle_c_log_unit_v(_p(x), _p(y), x.size); return y

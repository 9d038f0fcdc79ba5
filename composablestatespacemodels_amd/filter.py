"""Host-side mirror of the reference's particle-filter interface over libcssm_pf.

Reference (paths relative to src/main/scala/com/github/jonnylaw/model/):

* ``trait ParticleFilter[S]`` -- ParticleFilter.scala:96-167: ``initialiseState``, ``stepFilter``,
  ``llFilter``, ``filter``, ``filterStream``.
* ``Filter(mod, resample)`` :233-246, ``FilterInit(mod, resample, initState)`` :252-271,
  ``FilterLgcp(mod, resample, precision)`` :169-227.
* ``object ParticleFilter`` Reader entry points :321-361: ``filter``, ``filterInit``,
  ``filterLlState``, ``likelihood``; helpers ``effectiveSampleSize`` :431-434, ``mean`` :522-524.
* ``Resampling.systematicResampling`` -- Resampling.scala:63-72 (type ``Resample[A]``,
  package.scala:23).

Every computation happens in the HIP library; the particle cloud stays in HBM.  ``PfState`` is an
immutable value as in the reference (ParticleFilter.scala:32-37) except that ``particles`` is
fetched from the device on demand and is only available for the handle's CURRENT state.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .model import Data, Model, TimedObservation, split_data

_dp = C.POINTER(C.c_double)


def _p(a, ty=_dp):
    return a.ctypes.data_as(ty)


class NativePf:
    """Thin RAII wrapper of a ``cssm_pf*`` handle (AutoCloseable on the Scala side)."""

    def __init__(self, model: Model, n: int, seed: int = 20260101, device: int = 0, lgcp_precision: int = 0):
        self.lib = _abi.load_library()
        self._desc = model.descriptor(lgcp_precision)
        self._h = C.c_void_p()
        _abi.check(self.lib.cssm_pf_create(self._desc.ptr(), int(n), int(seed) & (2**64 - 1), int(device), C.byref(self._h)))
        self.n = int(n)
        self.d = int(self.lib.cssm_pf_dim(self._h))
        self.generation = 0
        # the same entry point bound once more with untyped pointers: run_more hands it raw array addresses
        self._ll_filter_more_raw = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double),
                                               C.c_void_p, C.c_void_p)(("cssm_pf_ll_filter_more", self.lib))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.cssm_pf_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, model: Model, lgcp_precision: int = 0):
        self._desc = model.descriptor(lgcp_precision)
        _abi.check(self.lib.cssm_pf_set_params(self._h, self._desc.ptr()))

    def reseed(self, seed: int):
        _abi.check(self.lib.cssm_pf_reseed(self._h, int(seed) & (2**64 - 1)))

    def init(self, t0: float):
        _abi.check(self.lib.cssm_pf_init(self._h, float(t0)))
        self.generation += 1

    def init_from(self, t0: float, state: Sequence[float]):
        s = np.ascontiguousarray(state, dtype=np.float64)
        if s.size != self.d:
            raise ValueError(f"initial state has {s.size} components, the model has {self.d}")
        _abi.check(self.lib.cssm_pf_init_from(self._h, float(t0), _p(s)))
        self.generation += 1

    def step(self, t: float, y: Optional[float], has_obs: Optional[bool] = None) -> Tuple[float, int]:
        if has_obs is None:
            has_obs = y is not None
        ll, ess = C.c_double(), C.c_int32()
        rc = self.lib.cssm_pf_step(self._h, float(t), 0.0 if y is None else float(y), 1 if has_obs else 0,
                                   C.byref(ll), C.byref(ess))
        self.generation += 1
        _abi.check(rc)
        return ll.value, ess.value

    def propagate(self, t: float, y: Optional[float], has_obs: Optional[bool] = None):
        """cssm_pf_propagate: stepFilter up to the weights (the caller resamples: see _FilterBase with a host function)."""
        if has_obs is None:
            has_obs = y is not None
        rc = self.lib.cssm_pf_propagate(self._h, float(t), 0.0 if y is None else float(y), 1 if has_obs else 0)
        self.generation += 1
        _abi.check(rc)

    def adopt(self, state: np.ndarray, ll: float, ess: int):
        s = np.ascontiguousarray(state, dtype=np.float64)
        if s.shape != (self.d, self.n):
            raise ValueError(f"the resampled cloud must be [{self.d}, {self.n}], got {s.shape}")
        _abi.check(self.lib.cssm_pf_adopt(self._h, _p(s), float(ll), int(ess)))
        self.generation += 1

    def run(self, t, y, has=None, want_path: bool = False):
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hp = None
        if has is not None:
            has = np.ascontiguousarray(has, dtype=np.uint8)
            hp = _p(has, C.POINTER(C.c_uint8))
        ll = C.c_double()
        ll_t = np.zeros(T)
        ess_t = np.zeros(T, dtype=np.int32)
        path = None
        if want_path:
            path = np.zeros((T + 1, self.d))
            rc = self.lib.cssm_pf_filter(self._h, _p(t), _p(y), hp, T, C.byref(ll), _p(ll_t),
                                         _p(ess_t, C.POINTER(C.c_int32)), _p(path))
        else:
            rc = self.lib.cssm_pf_ll_filter(self._h, _p(t), _p(y), hp, T, C.byref(ll), _p(ll_t),
                                            _p(ess_t, C.POINTER(C.c_int32)))
        self.generation += 1
        _abi.check(rc)
        return ll.value, ll_t, ess_t, path

    def run_more(self, t, y, has=None):
        """T MORE observations of the running filter (cssm_pf_ll_filter_more): no new cloud, the clock and the observation count
        go on; returns (ll accumulated since initialisation, ll_t, ess_t) of this call's observations."""
        # (a short continued leg pays for every microsecond here: no copies of arrays that already are what the ABI takes, raw
        #  addresses instead of typed ctypes pointers)
        if not (type(t) is np.ndarray and t.dtype == np.float64 and t.flags.c_contiguous):
            t = np.ascontiguousarray(t, dtype=np.float64)
        if not (type(y) is np.ndarray and y.dtype == np.float64 and y.flags.c_contiguous):
            y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hp = None
        if has is not None:
            if not (type(has) is np.ndarray and has.dtype == np.uint8 and has.flags.c_contiguous):
                has = np.ascontiguousarray(has, dtype=np.uint8)
            hp = has.ctypes.data
        ll = C.c_double()
        ll_t = np.empty(T)
        ess_t = np.empty(T, dtype=np.int32)
        rc = self._ll_filter_more_raw(self._h, t.ctypes.data, y.ctypes.data, hp, T, C.byref(ll), ll_t.ctypes.data, ess_t.ctypes.data)
        self.generation += 1
        _abi.check(rc)
        return ll.value, ll_t, ess_t

    def last_loop_ms(self) -> float:
        """Device time of the last batch call's per-observation kernels; needs set_option(9, 1) (CSSM_OPT_LOOP_EVENTS) before that call."""
        ms = C.c_float()
        _abi.check(self.lib.cssm_pf_last_loop_ms(self._h, C.byref(ms)))
        return ms.value

    def last_device_us(self) -> float:
        """Device time of the last continued batch call, first instruction of its first kernel to its closing kernel's results, from the GPU's
        constant clock (cssm_pf_last_device_us: no event packets on the queue)."""
        us = C.c_double()
        _abi.check(self.lib.cssm_pf_last_device_us(self._h, C.byref(us)))
        return us.value

    def stream_idle(self) -> bool:
        """Whether the runtime sees nothing queued or running on the handle's stream (cssm_pf_stream_idle; raises on a HIP error)."""
        r = self.lib.cssm_pf_stream_idle(self._h)
        if r < 0:
            _abi.check(r)
        return r == 1

    KERNELS = ("k_propagate", "k_tile_sums", "k_offspring", "k_reduce_units", "k_boundary_pack", "k_offspring_expand_spec", "collective")

    def set_option(self, option: int, value: int):
        _abi.check(self.lib.cssm_pf_set_option(self._h, int(option), int(value)))

    def profile(self, enable: bool):
        _abi.check(self.lib.cssm_pf_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """{kernel: (total_ms, launches)} accumulated since profile(True)."""
        ms = np.zeros(len(self.KERNELS))
        cnt = np.zeros(len(self.KERNELS), dtype=np.uint64)
        _abi.check(self.lib.cssm_pf_profile_read(self._h, _p(ms), _p(cnt, C.POINTER(C.c_uint64))))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.KERNELS)}

    def summary(self, interval: float = 0.975):
        """(state_mean[d], state_lower[d], state_upper[d], eta_of_mean, eta_lower, eta_upper) of the current cloud."""
        m, lo, hi = np.zeros(self.d), np.zeros(self.d), np.zeros(self.d)
        em, el, eu = C.c_double(), C.c_double(), C.c_double()
        _abi.check(self.lib.cssm_pf_summary(self._h, float(interval), _p(m), _p(lo), _p(hi), C.byref(em), C.byref(el), C.byref(eu)))
        return m, lo, hi, em.value, el.value, eu.value

    def interpolate(self, t, y, has=None, interval: float = 0.975, reference_pairing: bool = False):
        """cssm_pf_interpolate: (ll, mean[T+1,d], lower, upper, eta_of_mean[T+1], eta_lower, eta_upper)."""
        t = np.ascontiguousarray(t, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hp = None
        if has is not None:
            has = np.ascontiguousarray(has, dtype=np.uint8)
            hp = _p(has, C.POINTER(C.c_uint8))
        m, lo, hi = (np.zeros((T + 1, self.d)) for _ in range(3))
        em, el, eu = (np.zeros(T + 1) for _ in range(3))
        ll = C.c_double()
        rc = self.lib.cssm_pf_interpolate(self._h, _p(t), _p(y), hp, T, float(interval), 1 if reference_pairing else 0,
                                          C.byref(ll), _p(m), _p(lo), _p(hi), _p(em), _p(el), _p(eu))
        self.generation += 1
        _abi.check(rc)
        return ll.value, m, lo, hi, em, el, eu

    def particles(self) -> np.ndarray:
        out = np.zeros((self.d, self.n))
        _abi.check(self.lib.cssm_pf_get_particles(self._h, _p(out)))
        return out

    def proposed(self) -> np.ndarray:
        out = np.zeros((self.d, self.n))
        _abi.check(self.lib.cssm_pf_get_proposed(self._h, _p(out)))
        return out

    def weights(self):
        """(w1, c): the weights exp(min(w - c, 2^-20)) of the last weighted step and the level c they are relative to -- what the
        fused kernel keeps in place of the log-weights; None where the handle keeps log-weights (see ``logw``)."""
        out = np.zeros(self.n)
        level = C.c_double()
        rc = self.lib.cssm_pf_get_weights(self._h, _p(out), C.byref(level))
        if rc == -7:       # CSSM_ESTATE: log-weights are kept
            return None
        _abi.check(rc)
        return out, level.value

    def logw(self) -> np.ndarray:
        out = np.zeros(self.n)
        _abi.check(self.lib.cssm_pf_get_logw(self._h, _p(out)))
        return out

    def ancestors(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint32)
        _abi.check(self.lib.cssm_pf_get_ancestors(self._h, _p(out, C.POINTER(C.c_uint32))))
        return out


# --------------------------------------------------------------------------- Resample[A]
class NativePfBatch:
    """``cssm_pfb*``: B filters of one model structure advanced in lockstep, one launch per stage for all of them (the chains of a PMMH
    run, a pilot grid of parameters -- model/Streaming.scala:38-39).  ``filter(models, seeds, t, y, has)`` = B x ``NativePf.run(...,
    want_path=True)`` on handles reseeded with ``seeds[k]``, bit for bit."""

    def __init__(self, model: Model, n: int, chains: int, device: int = 0):
        self.lib = _abi.load_library()
        self._h = C.c_void_p()
        _abi.check(self.lib.cssm_pfb_create(model.descriptor().ptr(), int(n), int(chains), int(device), C.byref(self._h)))
        self.B, self.n = int(chains), int(n)
        self.d = int(self.lib.cssm_pf_dim(self.lib.cssm_pfb_chain(self._h, 0)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.cssm_pfb_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def filter(self, models: Sequence[Model], seeds: Sequence[int], t, y, has=None, want_path: bool = True):
        """(ll[B], path[B, T + 1, d] or None, rc[B]): rc[k] != 0 is chain k's own status (-5: its weights were unusable)."""
        if len(models) != self.B or len(seeds) != self.B:
            raise ValueError("one model and one seed per chain")
        t = np.ascontiguousarray(t, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        T = len(t)
        hp = None
        if has is not None:
            has = np.ascontiguousarray(has, dtype=np.uint8); hp = _p(has, C.POINTER(C.c_uint8))
        descs = [m.descriptor() for m in models]
        arr = (C.POINTER(_abi.ModelDesc) * self.B)(*[C.pointer(d.desc) for d in descs])
        sd = np.ascontiguousarray([int(x) & (2**64 - 1) for x in seeds], dtype=np.uint64)
        ll = np.zeros(self.B); rc = np.zeros(self.B, dtype=np.int32)
        path = np.zeros((self.B, T + 1, self.d)) if want_path else None
        _abi.check(self.lib.cssm_pfb_filter(self._h, arr, _p(sd, C.POINTER(C.c_uint64)), _p(t), _p(y), hp, T, _p(ll), _p(path) if want_path else None,
                                            _p(rc, C.POINTER(C.c_int))))
        return ll, path, rc

    def chain(self, k: int) -> "NativePf":
        """Chain k as a NativePf view (inspection only; the batch owns the handle)."""
        v = _PfView.__new__(_PfView)
        v.lib = self.lib; v._h = C.c_void_p(self.lib.cssm_pfb_chain(self._h, int(k))); v.n = self.n; v.d = self.d; v.generation = 0
        return v


class _PfView(NativePf):
    """A chain of a batch seen as a NativePf: the batch owns the handle, closing the view destroys nothing."""

    def close(self):
        self._h = C.c_void_p()

    __del__ = close


class Resampling:
    """``Resampling.systematicResampling`` (Resampling.scala:63-72) on the GPU.

    Called as a function it is the ``Resample[A]`` of package.scala:23: ``(samples, weights) ->
    samples`` where ``weights`` are the unnormalised ``w1 = exp(w - max)`` (ParticleFilter.scala:
    125-126).  Passed to ``Filter`` it only selects the native resampler -- the cloud never
    leaves the device.
    """

    @staticmethod
    def indentity(samples: Sequence, weights: Sequence[float]):
        """Resampling.indentity (sic), model/Resampling.scala:29: the samples as they are."""
        return samples

    @staticmethod
    def systematicAncestors(weights: Sequence[float], u: float, device: int = 0) -> np.ndarray:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        anc = np.zeros(len(w), dtype=np.uint32)
        _abi.check(_abi.load_library().cssm_resample_systematic(_p(w), len(w), float(u), _p(anc, C.POINTER(C.c_uint32)), device))
        return anc

    @staticmethod
    def ancestors(kind: int, weights: Sequence[float], u: float = 0.0, seed: int = 0, step: int = 0, device: int = 0) -> np.ndarray:
        """cssm_resample: the ancestor indices of resampler ``kind`` (0 systematic, 1 stratified, 2 multinomial)."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        anc = np.zeros(len(w), dtype=np.uint32)
        _abi.check(_abi.load_library().cssm_resample(int(kind), _p(w), len(w), float(u), int(seed), int(step),
                                                     _p(anc, C.POINTER(C.c_uint32)), device))
        return anc

    @staticmethod
    def _seeded(kind: int, particles: Sequence, weights: Sequence[float], seed: Optional[int], step: int):
        if len(particles) != len(weights):
            raise ValueError("particles and weights differ in length")
        if seed is None:   # the reference draws from unseeded global generators (:83, :93)
            seed = int(np.random.default_rng().integers(0, 2**63))
        return [particles[int(a)] for a in Resampling.ancestors(kind, weights, 0.0, seed, step)]

    @staticmethod
    def stratifiedResampling(particles: Sequence, weights: Sequence[float], seed: Optional[int] = None, step: int = 0):
        """Resampling.scala:78-86 as a ``Resample[A]``; as a ``Filter`` argument it selects the native stratified resampler."""
        return Resampling._seeded(1, particles, weights, seed, step)

    @staticmethod
    def multinomialResampling(particles: Sequence, weights: Sequence[float], seed: Optional[int] = None, step: int = 0):
        """Resampling.scala:92-96 as a ``Resample[A]``; as a ``Filter`` argument it selects the native multinomial resampler."""
        return Resampling._seeded(2, particles, weights, seed, step)

    @staticmethod
    def residualAncestors(weights: Sequence[float], seed: int = 0, step: int = 0, device: int = 0) -> np.ndarray:
        """cssm_resample_residual (an EXTENSION, see ``residualResampling``): the particle every slot takes."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        anc = np.zeros(len(w), dtype=np.uint32)
        _abi.check(_abi.load_library().cssm_resample_residual(_p(w), len(w), int(seed), int(step), _p(anc, C.POINTER(C.c_uint32)), device))
        return anc

    @staticmethod
    def residualResampling(particles: Sequence, weights: Sequence[float], seed: Optional[int] = None, step: int = 0):
        """EXTENSION.  The reference's residualResampling (Resampling.scala:130-146) cannot run as written: it hands ``Vector.range(1, m)``
        with n weights to the multinomial resampler, indexes the particles with the result (:144-145) and exp-normalises weights that
        are already exponentiated.  This is the resampler its scaladoc DESCRIBES (:124-129), as a host ``Resample[A]``: particle i
        appears floor(n w_i / sum w) times, the remaining slots are drawn by multinomial resampling on the residual weights
        (cssm_resample_residual).  ``weights`` are w1 = exp(w - max), as for the other resamplers.  As a ``Filter`` argument it runs
        through the host-resampler seam (cssm_pf_propagate / cssm_pf_adopt): there is no native in-filter kernel for it; the filter
        keys its draws by (the filter's seed, the number of observations it has resampled), so a seeded Filter reproduces.  Called
        directly with seed = None it takes a fresh seed, as the reference's unseeded generators would."""
        if len(particles) != len(weights):
            raise ValueError("particles and weights differ in length")
        if seed is None:
            seed = int(np.random.default_rng().integers(0, 2**63))
        return [particles[int(a)] for a in Resampling.residualAncestors(weights, seed, step)]

    @staticmethod
    def systematicResampling(particles: Sequence, weights: Sequence[float], u: Optional[float] = None):
        if len(particles) != len(weights):
            raise ValueError("particles and weights differ in length")
        if u is None:  # the reference draws from the unseeded global scala.util.Random (:66)
            u = float(np.random.default_rng().random())
        anc = Resampling.systematicAncestors(weights, u)
        return [particles[int(a)] for a in anc]


# --------------------------------------------------------------------------- PfState
@dataclass(frozen=True)
class PfState:  # ParticleFilter.scala:32-37
    t: float
    observation: Optional[float]
    ll: float
    ess: int
    _owner: Optional[NativePf] = field(default=None, repr=False, compare=False)
    _generation: int = field(default=-1, repr=False, compare=False)

    @property
    def particles(self) -> np.ndarray:
        """The resampled cloud, SoA ``[d, N]`` (row k = component k in Tree.flatten order)."""
        if self._owner is None or self._owner.generation != self._generation:
            raise RuntimeError("this PfState is not the handle's current state; its cloud was advanced on the device")
        return self._owner.particles()


@dataclass(frozen=True)
class CredibleInterval:  # ParticleFilter.scala:14
    lower: float
    upper: float


@dataclass(frozen=True)
class PfOut:  # ParticleFilter.scala:20-26
    time: float
    observation: Optional[float]
    eta: float
    etaIntervals: CredibleInterval
    state: np.ndarray
    stateIntervals: List[CredibleInterval]


@dataclass(frozen=True)
class StateSpace:  # Sde.scala:170
    time: float
    state: np.ndarray


# --------------------------------------------------------------------------- filters
class _FilterBase:
    lgcp_precision = 0

    def __init__(self, mod: Model, resample, n_particles: Optional[int] = None, seed: int = 20260101, device: int = 0):
        kinds = {Resampling.systematicResampling: 0, Resampling.stratifiedResampling: 1, Resampling.multinomialResampling: 2}
        if not callable(resample):
            raise TypeError("resample must be a Resample[A]: (samples, weights) -> samples")
        # one of the three native resamplers: the whole step stays on the device.  Any OTHER function is a host Resample[A]
        # (model/package.scala:23): the step is split at the resampler (cssm_pf_propagate / cssm_pf_adopt) and the function
        # is applied to the cloud's columns exactly as the reference applies it to its Vector[State] -- a parity path.
        self._resampler = kinds.get(resample, 0)
        self._host_resample = None if resample in kinds else resample
        self.mod = mod
        self.resample = resample
        self.seed = seed
        self.device = device
        self._pf: Optional[NativePf] = None
        self._host_step = 0
        if n_particles is not None:
            self._ensure(n_particles)

    def _ensure(self, n: int) -> NativePf:
        if self._pf is None or self._pf.n != n:
            if self._pf is not None:
                self._pf.close()
            self._pf = NativePf(self.mod, n, self.seed, self.device, self.lgcp_precision)
            if self._resampler:
                self._pf.set_option(2, self._resampler)   # CSSM_OPT_RESAMPLER
        return self._pf

    def _state(self, t, obs, ll, ess) -> PfState:
        return PfState(t, obs, ll, ess, self._pf, self._pf.generation)

    # ParticleFilter.scala:105-108
    def initialiseState(self, particles: int, t0: float) -> PfState:
        pf = self._ensure(particles)
        pf.init(t0)
        self._host_step = 0                                 # weighted observations resampled on the host since the cloud was drawn
        return self._state(t0, None, 0.0, particles)

    # ParticleFilter.scala:116-132 (FilterLgcp: :210-226)
    def stepFilter(self, s: PfState, y: TimedObservation) -> PfState:
        if s._owner is not self._pf or s._generation != self._pf.generation:
            raise RuntimeError("stepFilter must be applied to the filter's current PfState")
        if self._host_resample is not None:
            return self._step_with_host_resampler(s, y)
        ll, ess = self._pf.step(y.t, y.observation)
        return self._state(y.t, y.observation, ll, ess)

    def _step_with_host_resampler(self, s: PfState, y: TimedObservation) -> PfState:
        """stepFilter with the user's Resample[A], line by line (ParticleFilter.scala:116-132): the device propagates and
        weighs, the host rescales by the max, resamples with the given function and computes ll and ess."""
        pf = self._pf
        weighted = y.observation is not None or self.lgcp_precision > 0
        pf.propagate(y.t, y.observation)
        if not weighted:                                   # :121
            return self._state(y.t, y.observation, s.ll, s.ess)
        w = pf.logw()                                      # :123
        x1 = pf.proposed()
        mx = float(np.max(w))                              # :124
        w1 = np.exp(w - mx)                                # :125
        cols = [x1[:, i] for i in range(pf.n)]
        if self._host_resample is Resampling.residualResampling:
            # the residual extension draws its remainder from counter-based variates: keyed by the FILTER's seed and the observation's
            # index like the three native resamplers, so that Filter(seed = ...) reproduces (unkeyed it took a fresh seed per step)
            new = Resampling.residualResampling(cols, w1, seed=self.seed, step=self._host_step)
        else:
            new = self._host_resample(cols, w1)            # :126
        self._host_step += 1
        if len(new) != pf.n:
            raise ValueError("the resampler must return as many particles as it was given")
        ll = s.ll + mx + math.log(float(np.sum(w1)) / pf.n)   # :127
        nw = w1 / np.sum(w1)
        ess = int(math.floor(1.0 / float(np.sum(nw * nw))))   # :128
        pf.adopt(np.stack(new, axis=1), ll, ess)
        return self._state(y.t, y.observation, ll, ess)

    # ParticleFilter.scala:137-140
    def llFilter(self, data: Sequence[TimedObservation], n: int) -> float:
        if self._host_resample is not None:                # fold stepFilter over the data (:137-140)
            st = self.initialiseState(n, min(d.t for d in data))
            for d in data:
                st = self.stepFilter(st, d)
            return st.ll
        t, y, h = split_data(data)
        return self._ensure(n).run(t, y, h)[0]

    # ParticleFilter.scala:152-158
    def filter(self, data: Sequence[TimedObservation], particles: int) -> Tuple[float, List[StateSpace]]:
        if self._host_resample is not None:                # scan stepFilter, one uniformly picked particle per state (:152-158)
            rng = np.random.default_rng(self.seed)
            t0 = min(d.t for d in data)
            st = self.initialiseState(particles, t0)
            out = [StateSpace(t0, self._pf.particles()[:, int(rng.integers(particles))].copy())]
            for d in data:
                st = self.stepFilter(st, d)
                out.append(StateSpace(d.t, self._pf.particles()[:, int(rng.integers(particles))].copy()))
            return st.ll, out
        t, y, h = split_data(data)
        ll, _, _, path = self._ensure(particles).run(t, y, h, want_path=True)
        times = [float(np.min(t))] + [float(v) for v in t]
        return ll, [StateSpace(tt, path[i].copy()) for i, tt in enumerate(times)]

    # ParticleFilter.scala:163-166: Flow[Data].scan(init)(stepFilter) -- emits init, then one state per datum
    def filterStream(self, t0: float, particles: int) -> Callable[[Iterable[TimedObservation]], Iterator[PfState]]:
        def flow(source: Iterable[TimedObservation]) -> Iterator[PfState]:
            s = self.initialiseState(particles, t0)
            yield s
            for y in source:
                s = self.stepFilter(s, y)
                yield s
        return flow


class Filter(_FilterBase):
    """``Filter(mod, resample)``, ParticleFilter.scala:233-246."""


class FilterInit(_FilterBase):
    """``FilterInit(mod, resample, initState)``, ParticleFilter.scala:252-271."""

    def __init__(self, mod: Model, resample, initState: Sequence[float], **kw):
        super().__init__(mod, resample, **kw)
        self.initState = np.asarray(initState, dtype=np.float64)

    def initialiseState(self, particles: int, t0: float) -> PfState:
        pf = self._ensure(particles)
        pf.init_from(t0, self.initState)
        return self._state(t0, None, 0.0, particles)


class FilterLgcp(_FilterBase):
    """``FilterLgcp(mod, resample, precision)``, ParticleFilter.scala:169-227."""

    def __init__(self, mod: Model, resample, precision: int, **kw):
        self.lgcp_precision = int(precision)
        super().__init__(mod, resample, **kw)


class FilterInterpolate(_FilterBase):
    """``FilterInterpolate(mod, resample)``, ParticleFilter.scala:273-311: particles are whole paths and a weighted
    step resamples the paths.  The reference's only consumer (examples/Interpolate.scala:34-44) runs the stream to
    the end and summarises the LAST state's transposed paths; ``interpolate`` returns exactly that list of ``PfOut``
    (one per emitted state: the initial one and one per datum), formed on the device from the ancestor history.

    ``reference_pairing=True`` reproduces the example's ``zipped`` literally: the transposed paths run newest-first,
    so entry k carries the time/observation of index k but the states of index T-k; the default pairs them
    chronologically, which is what the example means to plot."""

    def interpolate(self, data: Sequence[TimedObservation], particles: int, interval: float = 0.975,
                    reference_pairing: bool = False) -> Tuple[float, List[PfOut]]:
        t, y, h = split_data(data)
        ll, m, lo, hi, em, el, eu = self._ensure(particles).interpolate(t, y, h, interval, reference_pairing)
        times = [float(np.min(t))] + [float(v) for v in t]
        obs = [None] + [d.observation for d in data]
        return ll, [PfOut(times[k], obs[k], float(em[k]), CredibleInterval(float(el[k]), float(eu[k])), m[k].copy(),
                          [CredibleInterval(float(a), float(b)) for a, b in zip(lo[k], hi[k])]) for k in range(len(times))]


class ParticleFilter:
    """``object ParticleFilter``: Reader-wrapped entry points, ParticleFilter.scala:321-361."""

    @staticmethod
    def filter(resample, t0: float, n: int, **kw):
        return lambda mod: Filter(mod, resample, **kw).filterStream(t0, n)

    @staticmethod
    def filterInit(resample, t0: float, n: int, initState, **kw):
        return lambda mod: FilterInit(mod, resample, initState, **kw).filterStream(t0, n)

    @staticmethod
    def interpolate(resample, t0: float, particles: int, **kw):   # :335-337 (t0 is re-derived from the data as minSink does)
        return lambda mod: (lambda data: FilterInterpolate(mod, resample, **kw).interpolate(list(data), particles))

    @staticmethod
    def filterLlState(data, resample, n: int, **kw):
        return lambda mod: Filter(mod, resample, **kw).filter(data, n)

    @staticmethod
    def likelihood(data, resample, n: int, **kw):
        return lambda mod: Filter(mod, resample, **kw).llFilter(data, n)

    @staticmethod
    def getIntervals(model: Model, s: "PfState") -> PfOut:
        """ParticleFilter.getIntervals (:415-424), evaluated on the device for the handle's current state."""
        m, lo, hi, em, el, eu = ParticleFilter._current(s).summary(0.975)
        return PfOut(s.t, s.observation, em, CredibleInterval(el, eu), m, [CredibleInterval(a, b) for a, b in zip(lo, hi)])

    @staticmethod
    def _current(s: "PfState") -> NativePf:
        if s._owner is None or s._owner.generation != s._generation:
            raise RuntimeError("the cloud summaries need the filter's current PfState (the cloud lives on the device)")
        return s._owner

    @staticmethod
    def meanState(s: "PfState") -> np.ndarray:
        """ParticleFilter.meanState (:475-477) of the state's cloud: per-component means, formed on the device."""
        return ParticleFilter._current(s).summary(0.975)[0]

    @staticmethod
    def getallCredibleIntervals(s: "PfState", interval: float) -> List[CredibleInterval]:
        """ParticleFilter.getallCredibleIntervals (:510-512, getCredibleInterval :488-502) of the state's cloud: per component
        the order statistics sorted(n - index - 1) and sorted(index - 1), index = floor(interval n), selected on the device."""
        _, lo, hi, _, _, _ = ParticleFilter._current(s).summary(float(interval))
        return [CredibleInterval(a, b) for a, b in zip(lo, hi)]

    @staticmethod
    def getOrderStatistic(samples: Sequence[float], interval: float) -> CredibleInterval:
        """ParticleFilter.getOrderStatistic (:455-460) for a vector the caller already holds on the host:
        CredibleInterval(sorted(n - index), sorted(index)), index = floor(n interval).  (The eta intervals of a cloud come
        from getIntervals, which selects them on the device.)"""
        a = np.sort(np.asarray(samples, dtype=np.float64))
        index = int(np.floor(a.size * interval))
        return CredibleInterval(float(a[a.size - index]), float(a[index]))

    @staticmethod
    def effectiveSampleSize(weights: Sequence[float]) -> int:  # :431-434 (host helper, tiny inputs)
        w = np.asarray(weights, dtype=np.float64)
        nw = w / w.sum()
        return int(np.floor(1.0 / np.sum(nw * nw)))

    @staticmethod
    def mean(s: Sequence[float]) -> float:  # :522-524
        return float(np.sum(s)) / len(s)

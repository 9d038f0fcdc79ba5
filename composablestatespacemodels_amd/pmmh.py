"""Host-side mirror of the reference's PMMH surface (model/PMMH.scala, model/MarkovChain.scala).

The MCMC chain is sequential and stays on the host (BASELINE.json north star); each iteration's
work is one native filter run.  Two entry points:

* ``MetropolisHastings.pmmhState(initP, proposal, logTransition, prior)(pf)`` -- the generic
  reference signature (PMMH.scala:161-167): ``pf`` is a ``BootstrapFilter`` (package.scala:24), i.e.
  a callable ``params -> (ll, path)``; ``proposal(params, rng)`` draws a proposal.  Returns an
  iterator of ``MetropState`` (the stream drops the initial state, PMMH.scala:97).
* ``pmmh_native`` -- the wiring of examples/DetermineParameters.scala:58-80 (``perturb(delta)``
  proposal, symmetric transition, flat prior) executed by ``cssm_pmmh_run`` inside the library with
  the contract's Philox streams, so a CPU restatement reproduces the chain bit for bit.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Iterator, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .filter import NativePf, Resampling
from .model import Parameters, UnparamModel, split_data


@dataclass
class MetropState:  # PMMH.scala:10-14
    ll: float
    params: Parameters
    sde: object
    accepted: int


@dataclass
class ParamsState:  # PMMH.scala:17 (what `params` emits: the chain without the sampled state)
    ll: float
    params: Parameters
    accepted: int


class ParametersProposal:
    """``Parameters.perturb(delta)`` (Parameters.scala:65-67): independent Gaussian(theta_k, sqrt(delta))
    on every stored scalar, scale included."""

    def __init__(self, delta: float):
        self.sd = math.sqrt(delta)

    def __call__(self, p: Parameters, rng: np.random.Generator) -> Parameters:
        theta = np.asarray(p.flattenParams())
        return p.withFlat(theta + self.sd * rng.standard_normal(theta.size))


class MetropolisHastings:
    @staticmethod
    def mhStep(s: MetropState, proposal, logTransition, prior, pf, rng) -> MetropState:
        """PMMH.scala:68-81; the stored ll of the current state is reused, not re-estimated (:63-66)."""
        prop = proposal(s.params, rng)
        ll, path = pf(prop)
        a = ll + logTransition(prop, s.params) + prior(prop) - logTransition(s.params, prop) - s.ll - prior(s.params)
        u = rng.random()
        if math.log(u) < a if u > 0.0 else True:
            return MetropState(ll, prop, path[-1] if len(path) else None, s.accepted + 1)
        return s

    @staticmethod
    def pmmhState(initP: Parameters, proposal, logTransition, prior):
        def run(pf: Callable[[Parameters], Tuple[float, Sequence]], rng: Optional[np.random.Generator] = None,
                iters: Optional[int] = None) -> Iterator[MetropState]:
            rng = rng or np.random.default_rng()
            s = MetropState(-1e99, initP, None, 0)      # PMMH.scala:121
            k = 0
            while iters is None or k < iters:
                s = MetropolisHastings.mhStep(s, proposal, logTransition, prior, pf, rng)
                k += 1
                yield s
        return run


    @staticmethod
    def approxMhStep(s: MetropState, proposal, logTransition, prior, pf, rng) -> MetropState:
        """ApproxPMMH.mhStep (PMMH.scala:138-152): the likelihood of the CURRENT parameters is estimated again in every
        step (two filter runs per iteration: the proposal's first, then the current one's), and a rejected step carries
        that fresh estimate and its last sampled state."""
        prop = proposal(s.params, rng)
        ll, path = pf(prop)
        old_ll, old_path = pf(s.params)
        a = ll + logTransition(prop, s.params) + prior(prop) - logTransition(s.params, prop) - old_ll - prior(s.params)
        u = rng.random()
        if math.log(u) < a if u > 0.0 else True:
            return MetropState(ll, prop, path[-1] if len(path) else None, s.accepted + 1)
        return MetropState(old_ll, s.params, old_path[-1] if len(old_path) else None, s.accepted)

    @staticmethod
    def approxPmmh(initP: Parameters, proposal, logTransition, prior):
        """MetropolisHastings.approxPmmh (PMMH.scala:169-175): as pmmhState with ApproxPMMH's step."""
        def run(pf: Callable[[Parameters], Tuple[float, Sequence]], rng: Optional[np.random.Generator] = None,
                iters: Optional[int] = None) -> Iterator[MetropState]:
            rng = rng or np.random.default_rng()
            s = MetropState(-1e99, initP, None, 0)      # PMMH.scala:135
            k = 0
            while iters is None or k < iters:
                s = MetropolisHastings.approxMhStep(s, proposal, logTransition, prior, pf, rng)
                k += 1
                yield s
        return run

    @staticmethod
    def params(chain: Iterator[MetropState]) -> Iterator[ParamsState]:
        """MetropolisHastings.params (PMMH.scala:89-93): the chain's (ll, params, accepted)."""
        for s in chain:
            yield ParamsState(s.ll, s.params, s.accepted)

    @staticmethod
    def pmmhStep(pos: Callable, proposal: Callable, s: Tuple[float, object], rng) -> Tuple[float, object]:
        """MetropolisHastings.pmmhStep (PMMH.scala:177-191): Metropolis step on a (log-posterior, parameter) pair with a
        symmetric proposal; `pos` returns the log-posterior estimate of a proposal."""
        prop = proposal(s[1], rng)
        ll = pos(prop)
        u = rng.random()
        return (ll, prop) if (u <= 0.0 or math.log(u) < ll - s[0]) else s


def bootstrap_filter(unparam: UnparamModel, data, n: int, seed: int = 20260101, device: int = 0):
    """``Reader { p => Filter(model.run(p).get, resample).filter(data, n) }`` on one native handle
    (examples/DetermineParameters.scala:70-72): re-parameterised per call, never re-allocated."""
    t, y, h = split_data(data)
    state = {"pf": None, "calls": 0}

    def pf(p: Parameters):
        model = unparam.run(p)
        if state["pf"] is None:
            state["pf"] = NativePf(model, n, seed, device)
        else:
            state["pf"].set_params(model)
        state["pf"].reseed(state["pf"].lib.cssm_pf_run_key(seed & (2**64 - 1), 1 + state["calls"]))   # a PRF of (seed, call), never seed + call
        state["calls"] += 1
        ll, _, _, path = state["pf"].run(t, y, h, want_path=True)
        return ll, path
    return pf


def pmmh_native(unparam: UnparamModel, init: Parameters, data, n: int, delta: float, iters: int,
                seed: int = 20260101, device: int = 0):
    """Returns (ll[iters], theta[iters, n_theta], accepted[iters], last_state[iters, d])."""
    t, y, h = split_data(data)
    model = unparam.run(init)
    pf = NativePf(model, n, seed, device)
    theta0 = np.ascontiguousarray(init.flattenParams(), dtype=np.float64)
    nt = theta0.size
    ll = np.zeros(iters); th = np.zeros((iters, nt)); acc = np.zeros(iters, dtype=np.int32); last = np.zeros((iters, pf.d))
    dp = C.POINTER(C.c_double)
    desc = model.descriptor()
    rc = pf.lib.cssm_pmmh_run(pf._h, desc.ptr(), theta0.ctypes.data_as(dp), nt, float(delta), t.ctypes.data_as(dp),
                              y.ctypes.data_as(dp), h.ctypes.data_as(C.POINTER(C.c_uint8)), len(t), int(seed), int(iters),
                              ll.ctypes.data_as(dp), th.ctypes.data_as(dp), acc.ctypes.data_as(C.POINTER(C.c_int32)),
                              last.ctypes.data_as(dp))
    pf.close()
    _abi.check(rc)
    return ll, th, acc, last


def pmmh_native_batched(unparam: UnparamModel, inits: Sequence[Parameters], data, n: int, delta: float, iters: int,
                        seeds: Sequence[int], device: int = 0):
    """``len(inits)`` chains in lockstep (``cssm_pmmh_run_batched``): chain k is ``pmmh_native(unparam, inits[k], ..., seed=seeds[k])`` bit
    for bit -- what examples/DetermineParameters.scala:68-69 runs as two chains under ``mapAsync(2)`` -- but every iteration's filters
    of all chains run as ONE batch on the GPU.  Returns (ll[B, iters], theta[B, iters, n_theta], accepted[B, iters], last_state[B, iters, d])."""
    t, y, h = split_data(data)
    B = len(inits)
    if B < 1 or len(seeds) != B:
        raise ValueError("one seed per chain")
    model = unparam.run(inits[0])
    desc = model.descriptor()
    lib = _abi.load_library()
    hb = C.c_void_p()
    _abi.check(lib.cssm_pfb_create(desc.ptr(), int(n), B, int(device), C.byref(hb)))
    try:
        d = int(lib.cssm_pf_dim(lib.cssm_pfb_chain(hb, 0)))
        theta0 = np.ascontiguousarray([p.flattenParams() for p in inits], dtype=np.float64)
        nt = theta0.shape[1]
        ll = np.zeros((B, iters)); th = np.zeros((B, iters, nt)); acc = np.zeros((B, iters), dtype=np.int32); last = np.zeros((B, iters, d))
        sd = np.ascontiguousarray([int(x) & (2**64 - 1) for x in seeds], dtype=np.uint64)
        dp = C.POINTER(C.c_double)
        _abi.check(lib.cssm_pmmh_run_batched(hb, desc.ptr(), theta0.ctypes.data_as(dp), nt, float(delta), t.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                             h.ctypes.data_as(C.POINTER(C.c_uint8)), len(t), sd.ctypes.data_as(C.POINTER(C.c_uint64)), int(iters),
                                             ll.ctypes.data_as(dp), th.ctypes.data_as(dp), acc.ctypes.data_as(C.POINTER(C.c_int32)), last.ctypes.data_as(dp)))
    finally:
        lib.cssm_pfb_destroy(hb)
    return ll, th, acc, last


def pmmh_native_speculative(unparam: UnparamModel, init: Parameters, data, n: int, delta: float, iters: int,
                            seed: int = 20260101, device: int = 0):
    """``pmmh_native`` -- the same chain, bit for bit -- at two iterations per batch of three filters (``cssm_pmmh_run_speculative``):
    iteration i's proposal and both candidates for iteration i + 1 (proposed from it, and from the current parameters) are filtered
    together; the proposals and filter keys are functions of (seed, iteration, parameters) alone.  For clouds that leave the GPU
    mostly idle (BASELINE configs[4]: N = 100 000).  Returns (ll[iters], theta[iters, n_theta], accepted[iters], last_state[iters, d])."""
    t, y, h = split_data(data)
    model = unparam.run(init)
    desc = model.descriptor()
    lib = _abi.load_library()
    hb = C.c_void_p()
    _abi.check(lib.cssm_pfb_create(desc.ptr(), int(n), 3, int(device), C.byref(hb)))
    try:
        d = int(lib.cssm_pf_dim(lib.cssm_pfb_chain(hb, 0)))
        theta0 = np.ascontiguousarray(init.flattenParams(), dtype=np.float64)
        nt = theta0.size
        ll = np.zeros(iters); th = np.zeros((iters, nt)); acc = np.zeros(iters, dtype=np.int32); last = np.zeros((iters, d))
        dp = C.POINTER(C.c_double)
        _abi.check(lib.cssm_pmmh_run_speculative(hb, desc.ptr(), theta0.ctypes.data_as(dp), nt, float(delta), t.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                                 h.ctypes.data_as(C.POINTER(C.c_uint8)), len(t), int(seed) & (2**64 - 1), int(iters),
                                                 ll.ctypes.data_as(dp), th.ctypes.data_as(dp), acc.ctypes.data_as(C.POINTER(C.c_int32)), last.ctypes.data_as(dp)))
    finally:
        lib.cssm_pfb_destroy(hb)
    return ll, th, acc, last

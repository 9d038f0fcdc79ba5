"""Host-side mirror of the reference's model-definition surface, just enough of it to describe a
composed POMP model to the native filter.

Names and argument meaning follow the reference (paths relative to
src/main/scala/com/github/jonnylaw/model/):

* ``SdeParameter.brownianParameter / genBrownianParameter / ouParameter`` and their
  ``*Unconstrained`` twins -- SdeParameters.scala:176-205.  The constrained constructors store
  ``log c0``, ``log sigma`` and, for OU, ``logistic(phi)`` (sic, :204); the SDE constructor applies
  ``logistic`` again (Sde.scala:136).  That double-logistic is reproduced, not repaired.
* ``Parameters(scale, sdeParam)`` leaves, composition with ``|`` for the reference's ``|+|`` --
  Parameters.scala:14-23, Tree.scala:18-20 (left-nested branches).
* ``Sde.brownianMotion / genBrownianMotion / ouProcess`` -- Sde.scala:181-202;
  ``Model.poisson / linear / seasonal / lgcp`` and ``|`` composition -- Model.scala:44-136.
  ``.run(params)`` is ``ReaderT.run``: it raises where the reference returns ``Failure``.

Nothing here computes on particles: the parameterised ``Model`` only knows how to flatten itself
into the ``cssm_model_desc`` of include/cssm_pf.h.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

from . import _abi


def _vec(x) -> List[float]:
    if isinstance(x, (int, float)):
        return [float(x)]
    return [float(v) for v in x]


def logistic(x: float) -> float:
    """SdeParameter.logistic, SdeParameters.scala:214-216."""
    return 1.0 / (1.0 + math.exp(-x))


def logit(p: float) -> float:
    """SdeParameter.logit, SdeParameters.scala:210-212."""
    return math.log(p) - math.log(1 - p)


# --------------------------------------------------------------------------- SDE parameters
@dataclass
class BrownianParameter:
    m0: List[float]
    c0: List[float]
    sigma: List[float]
    kind = _abi.SDE_BROWNIAN

    def flatten(self) -> List[float]:  # SdeParameters.scala:111
        return self.m0 + self.c0 + self.sigma

    def unflatten(self, v: Sequence[float]) -> "BrownianParameter":
        a, b = len(self.m0), len(self.m0) + len(self.c0)
        return BrownianParameter(list(v[:a]), list(v[a:b]), list(v[b:]))


@dataclass
class GenBrownianParameter:
    m0: List[float]
    c0: List[float]
    mu: List[float]
    sigma: List[float]
    kind = _abi.SDE_GEN_BROWNIAN

    def flatten(self) -> List[float]:  # SdeParameters.scala:73
        return self.m0 + self.c0 + self.mu + self.sigma

    def unflatten(self, v):
        a = len(self.m0); b = a + len(self.c0); c = b + len(self.mu)
        return GenBrownianParameter(list(v[:a]), list(v[a:b]), list(v[b:c]), list(v[c:]))


@dataclass
class OuParameter:
    m0: List[float]
    c0: List[float]
    phi: List[float]
    mu: List[float]
    sigma: List[float]
    kind = _abi.SDE_OU

    def flatten(self) -> List[float]:  # SdeParameters.scala:153-154
        return self.m0 + self.c0 + self.phi + self.mu + self.sigma

    def unflatten(self, v):
        a = len(self.m0); b = a + len(self.c0); c = b + len(self.phi); d = c + len(self.mu)
        return OuParameter(list(v[:a]), list(v[a:b]), list(v[b:c]), list(v[c:d]), list(v[d:]))


@dataclass
class EulerAffineParameter:
    """A user-defined Sde served by the trait's default Euler-Maruyama step (Sde.scala:23-43) with
    drift a + b*x and constant diagonal diffusion g; m0 and log c0 as for the built-ins."""
    m0: List[float]
    c0: List[float]
    b: List[float]
    a: List[float]
    g: List[float]
    kind = _abi.SDE_EULER_AFFINE

    def flatten(self):
        return self.m0 + self.c0 + self.b + self.a + self.g

    def unflatten(self, v):
        i = len(self.m0); j = i + len(self.c0); k = j + len(self.b); l = k + len(self.a)
        return EulerAffineParameter(list(v[:i]), list(v[i:j]), list(v[j:k]), list(v[k:l]), list(v[l:]))


SdeParam = Union[BrownianParameter, GenBrownianParameter, OuParameter, EulerAffineParameter]


class SdeParameter:
    """Smart constructors, SdeParameters.scala:176-205."""

    @staticmethod
    def brownianParameterUnconstrained(m0, c0, sigma) -> BrownianParameter:
        return BrownianParameter(_vec(m0), _vec(c0), _vec(sigma))

    @staticmethod
    def genBrownianParameterUnconstrained(m0, c0, mu, sigma) -> GenBrownianParameter:
        return GenBrownianParameter(_vec(m0), _vec(c0), _vec(mu), _vec(sigma))

    @staticmethod
    def ouParameterUnconstrained(m0, c0, phi, mu, sigma) -> OuParameter:
        return OuParameter(_vec(m0), _vec(c0), _vec(phi), _vec(mu), _vec(sigma))

    @staticmethod
    def brownianParameter(m0, c0, sigma) -> BrownianParameter:  # :197-200
        return BrownianParameter(_vec(m0), [math.log(c) for c in _vec(c0)], [math.log(s) for s in _vec(sigma)])

    @staticmethod
    def genBrownianParameter(m0, c0, mu, sigma) -> GenBrownianParameter:  # :192-195
        return GenBrownianParameter(_vec(m0), [math.log(c) for c in _vec(c0)], _vec(mu),
                                    [math.log(s) for s in _vec(sigma)])

    @staticmethod
    def ouParameter(m0, c0, phi, mu, sigma) -> OuParameter:  # :202-205 -- phi.map(logistic), sic
        return OuParameter(_vec(m0), [math.log(c) for c in _vec(c0)], [logistic(p) for p in _vec(phi)],
                           _vec(mu), [math.log(s) for s in _vec(sigma)])

    @staticmethod
    def eulerAffineParameter(m0, c0, b, a, g) -> EulerAffineParameter:
        return EulerAffineParameter(_vec(m0), [math.log(c) for c in _vec(c0)], _vec(b), _vec(a), _vec(g))


# --------------------------------------------------------------------------- parameter tree
@dataclass
class ParamNode:  # Parameters.scala:14
    scale: Optional[float]
    sdeParam: SdeParam


class Parameters:
    """``Tree[ParamNode]`` restricted to what composition builds: a left-nested list of leaves."""

    def __init__(self, leaves: List[ParamNode]):
        self.leaves = leaves

    @staticmethod
    def apply(scale: Optional[float], sdeParam: SdeParam) -> "Parameters":  # Parameters.scala:20-22
        return Parameters([ParamNode(scale, sdeParam)])

    leaf = apply

    def __or__(self, other: "Parameters") -> "Parameters":  # |+| on trees, Tree.scala:18-20
        return Parameters(self.leaves + other.leaves)

    def flattenParams(self) -> List[float]:  # Parameters.scala:88-95
        out: List[float] = []
        for n in self.leaves:
            if n.scale is not None:
                out.append(n.scale)
            out.extend(n.sdeParam.flatten())
        return out

    def paramSize(self) -> int:
        return len(self.flattenParams())

    def withFlat(self, theta: Sequence[float]) -> "Parameters":
        """Inverse of flattenParams (same tree shape, new values)."""
        out, i = [], 0
        for n in self.leaves:
            scale = None
            if n.scale is not None:
                scale = float(theta[i]); i += 1
            k = len(n.sdeParam.flatten())
            out.append(ParamNode(scale, n.sdeParam.unflatten([float(v) for v in theta[i:i + k]])))
            i += k
        if i != len(theta):
            raise ValueError("theta has the wrong length for this parameter tree")
        return Parameters(out)

    def add(self, delta: Sequence[float]) -> "Parameters":  # Addable[Parameters], Parameters.scala:97-104
        cur = self.flattenParams()
        return self.withFlat([a + b for a, b in zip(cur, delta)])


# --------------------------------------------------------------------------- SDEs
@dataclass
class UnparamSde:
    kind: int
    dimension: int

    def run(self, p: SdeParam) -> "SdeInstance":
        if p.kind != self.kind:  # Failure(...) at Sde.scala:183,188,201
            raise ValueError(f"Incorrect parameters supplied to SDE kind {self.kind}, received {type(p).__name__}")
        if self.dimension < 1:
            raise ValueError("SDE dimension must be >= 1")
        return SdeInstance(self.kind, self.dimension, p)


@dataclass
class SdeInstance:
    kind: int
    dimension: int
    params: SdeParam


class Sde:
    @staticmethod
    def brownianMotion(dimension: int) -> UnparamSde:  # Sde.scala:186-189
        return UnparamSde(_abi.SDE_BROWNIAN, dimension)

    @staticmethod
    def genBrownianMotion(dimension: int) -> UnparamSde:  # Sde.scala:181-184
        return UnparamSde(_abi.SDE_GEN_BROWNIAN, dimension)

    @staticmethod
    def ouProcess(dimension: int) -> UnparamSde:  # Sde.scala:199-202
        return UnparamSde(_abi.SDE_OU, dimension)

    @staticmethod
    def eulerAffine(dimension: int) -> UnparamSde:  # a user Sde on the trait default, Sde.scala:23-43
        return UnparamSde(_abi.SDE_EULER_AFFINE, dimension)


# --------------------------------------------------------------------------- models
@dataclass
class _LeafSpec:
    obs: str  # 'poisson' | 'linear' | 'seasonal' | 'lgcp' | 'negbin' | 'zip' | 'bernoulli' | 'studentt' | 'beta'
    sde: UnparamSde
    period: int = 0
    harmonics: int = 0
    df: int = 0


class UnparamModel:
    """``ReaderT[Try, Parameters, Model]``; ``a | b`` is ``a |+| b`` (Model.scala:97-136)."""

    def __init__(self, specs: List[_LeafSpec]):
        self.specs = specs

    def __or__(self, other: "UnparamModel") -> "UnparamModel":
        return UnparamModel(self.specs + other.specs)

    def run(self, p: Parameters) -> "Model":
        if len(p.leaves) != len(self.specs):
            # "Can't build model from branch parameter" / "... composed model from Leaf Parameter"
            raise ValueError("parameter tree shape does not match the composed model (Model.scala:47,134)")
        leaves = []
        for spec, node in zip(self.specs, p.leaves):
            sde = spec.sde.run(node.sdeParam)
            if spec.obs == "seasonal" and sde.dimension != 2 * spec.harmonics:
                raise ValueError("seasonal model needs an SDE of dimension 2*harmonics (Model.scala:217-225)")
            leaves.append((spec, node, sde))
        return Model(leaves)


class Model:
    """A parameterised (possibly composed) model.  Only the LEFTMOST leaf supplies the observation
    distribution (Model.scala:118-120,132)."""

    def __init__(self, leaves):
        self.leaves = leaves
        first = leaves[0][0].obs
        if first == "poisson":
            self.obs_kind = _abi.OBS_POISSON
        elif first in ("linear", "seasonal"):
            self.obs_kind = _abi.OBS_GAUSSIAN
            if leaves[0][1].scale is None:  # Model.scala:214,250
                raise ValueError("Must provide SD parameter for LinearModel / SeasonalModel")
        elif first == "lgcp":
            self.obs_kind = _abi.OBS_LGCP
        elif first in ("negbin", "zip", "studentt"):
            self.obs_kind = {"negbin": _abi.OBS_NEGBIN, "zip": _abi.OBS_ZIP, "studentt": _abi.OBS_STUDENT_T}[first]
            if leaves[0][1].scale is None:  # Model.scala:150,179,294
                raise ValueError("No scale parameter provided to the observation model")
        elif first in ("bernoulli", "beta"):
            self.obs_kind = _abi.OBS_BERNOULLI if first == "bernoulli" else _abi.OBS_BETA
        else:
            raise ValueError(f"unknown observation model {first}")
        self.obs_df = leaves[0][0].df
        self.dimension = sum(l[2].dimension for l in leaves)
        if self.dimension > _abi.MAX_DIM or len(leaves) > _abi.MAX_LEAVES:
            raise ValueError("model too large for the native filter (CSSM_MAX_DIM / CSSM_MAX_LEAVES)")

    # -- static constructors, Model.scala:44-95
    @staticmethod
    def poisson(sde: UnparamSde) -> UnparamModel:
        return UnparamModel([_LeafSpec("poisson", sde)])

    @staticmethod
    def linear(sde: UnparamSde) -> UnparamModel:
        return UnparamModel([_LeafSpec("linear", sde)])

    @staticmethod
    def seasonal(period: int, harmonics: int, sde: UnparamSde) -> UnparamModel:
        return UnparamModel([_LeafSpec("seasonal", sde, period, harmonics)])

    @staticmethod
    def lgcp(sde: UnparamSde) -> UnparamModel:
        return UnparamModel([_LeafSpec("lgcp", sde)])

    @staticmethod
    def negativeBinomial(sde: UnparamSde) -> UnparamModel:  # Model.scala:87-89
        return UnparamModel([_LeafSpec("negbin", sde)])

    @staticmethod
    def zeroInflatedPoisson(sde: UnparamSde) -> UnparamModel:  # Model.scala:91-94
        return UnparamModel([_LeafSpec("zip", sde)])

    @staticmethod
    def bernoulli(sde: UnparamSde) -> UnparamModel:  # Model.scala:77-80
        return UnparamModel([_LeafSpec("bernoulli", sde)])

    @staticmethod
    def studentsT(sde: UnparamSde, df: int) -> UnparamModel:  # Model.scala:72-75
        return UnparamModel([_LeafSpec("studentt", sde, df=int(df))])

    @staticmethod
    def beta(sde: UnparamSde) -> UnparamModel:  # Model.scala:50-55
        return UnparamModel([_LeafSpec("beta", sde)])

    def parameters(self) -> Parameters:
        return Parameters([l[1] for l in self.leaves])

    def descriptor(self, lgcp_precision: int = 0) -> "Descriptor":
        return Descriptor(self, lgcp_precision)


class Descriptor:
    """Owns the ctypes arrays behind a ``cssm_model_desc`` (keeps them alive)."""

    def __init__(self, model: Model, lgcp_precision: int = 0):
        n = len(model.leaves)
        self._keep = []
        self.leaf_array = (_abi.LeafDesc * n)()
        for i, (spec, node, sde) in enumerate(model.leaves):
            L = self.leaf_array[i]
            L.sde_kind = sde.kind
            L.dim = sde.dimension
            L.f_kind = _abi.F_SEASONAL if spec.obs == "seasonal" else _abi.F_FIRST
            L.period = spec.period
            L.harmonics = spec.harmonics
            L.has_scale = 0 if node.scale is None else 1
            L.scale = 0.0 if node.scale is None else node.scale
            p = sde.params

            def arr(v):
                a = (C.c_double * len(v))(*v)
                self._keep.append(a)
                return C.cast(a, C.POINTER(C.c_double)), len(v)

            L.m0, L.n_m0 = arr(p.m0)
            L.c0, L.n_c0 = arr(p.c0)
            if isinstance(p, EulerAffineParameter):
                L.mu, L.n_mu = arr(p.a)
                L.phi, L.n_phi = arr(p.b)
                L.sigma, L.n_sigma = arr(p.g)
            else:
                if hasattr(p, "mu"):
                    L.mu, L.n_mu = arr(p.mu)
                if hasattr(p, "phi"):
                    L.phi, L.n_phi = arr(p.phi)
                L.sigma, L.n_sigma = arr(p.sigma)
        self.desc = _abi.ModelDesc()
        self.desc.n_leaves = n
        self.desc.obs_kind = model.obs_kind
        self.desc.lgcp_precision = lgcp_precision
        self.desc.obs_df = getattr(model, "obs_df", 0)
        self.desc.leaves = C.cast(self.leaf_array, C.POINTER(_abi.LeafDesc))

    def ptr(self):
        return C.byref(self.desc)


# --------------------------------------------------------------------------- data
@dataclass
class TimedObservation:  # Data.scala:18-21
    t: float
    observation: Optional[float]


Data = TimedObservation


def split_data(data: Sequence[TimedObservation]):
    """(times, values, has_obs) arrays in the order given (no sort: ParticleFilter.scala:139)."""
    import numpy as np
    t = np.array([d.t for d in data], dtype=np.float64)
    y = np.array([0.0 if d.observation is None else d.observation for d in data], dtype=np.float64)
    h = np.array([0 if d.observation is None else 1 for d in data], dtype=np.uint8)
    return t, y, h

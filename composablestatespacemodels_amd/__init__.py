"""composablestatespacemodels_amd -- MI355X-native bootstrap particle filter behind the
ParticleFilter / Resample / BootstrapFilter seams of jonnylaw/ComposableStateSpaceModels.

The compute path is libcssm_pf.so (hand-written HIP for gfx950, C ABI in include/cssm_pf.h).
This package is the thin host-side mirror of the reference's interface for that path.
"""
from . import _abi
from ._abi import CssmError, load_library
from .model import (Data, Model, Parameters, ParamNode, Sde, SdeParameter, TimedObservation,
                    UnparamModel, UnparamSde, logistic, logit)

__all__ = ["_abi", "CssmError", "load_library", "Data", "Model", "Parameters", "ParamNode", "Sde",
           "SdeParameter", "TimedObservation", "UnparamModel", "UnparamSde", "logistic", "logit"]

"""Wire / disk formats at the edges of the hot path: read the observation files the
reference simulates, write posterior and filter output its R scripts plot.  No arithmetic on particles.

Reference (paths relative to src/main/scala/com/github/jonnylaw/model/):

* CSV observations ``t, y`` -- ``DataFromFile`` Data.scala:252-260 (second column empty = missing);
  the ``Show[Data]`` writer CsvFormat.scala:16-22 prints ``NA`` for a missing value, which the reference's
  own reader would choke on (``"NA".toDouble``); this reader accepts both.
* JSON-lines observations ``{"t": .., "observation": ..}`` -- ``DataFromJson`` Data.scala:265-272 with
  ``jsonFormat2(TimedObservation)`` jsonFormats.scala:125 (spray omits a ``None`` field).
* ``State`` as a JSON array of leaves ``[{"value": [..]}, ..]`` -- jsonFormats.scala:80-101.
* ``Parameters`` as a JSON array of ``{"scale": .., "sdeParam": {m0, c0, [mu], [phi], sigma}}`` where the SDE kind
  is recognised by the NUMBER of fields (3 Brownian, 4 GenBrownian, 5 OU) -- jsonFormats.scala:29-77.
* ``MetropState`` ``{"ll", "params", "sde": {"time", "state"}, "accepted"}`` -- jsonFormats.scala:121-122,
  written one per line by Streaming.pmmhToJson (Streaming.scala:42-58).
* CSV lines of ``Parameters`` / ``MetropState`` / ``PfOut`` -- CsvFormat.scala:33-47,53-61,75-83.
"""
from __future__ import annotations

import json
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

from .filter import CredibleInterval, PfOut, StateSpace
from .model import (BrownianParameter, GenBrownianParameter, OuParameter, ParamNode, Parameters, TimedObservation)
from .pmmh import MetropState


# --------------------------------------------------------------------------- observations
def read_csv_observations(path: str) -> List[TimedObservation]:
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            d = [c.strip() for c in line.split(",")]
            y = None if len(d) < 2 or d[1] in ("", "NA") else float(d[1])
            out.append(TimedObservation(float(d[0]), y))
    return out


def write_csv_observations(path: str, data: Iterable[TimedObservation]) -> None:
    with open(path, "w") as f:
        for d in data:  # Show[Data], CsvFormat.scala:18
            f.write(f"{d.t}, {'NA' if d.observation is None else d.observation}\n")


def observation_to_json(d: TimedObservation) -> str:
    o = {"t": d.t}
    if d.observation is not None:
        o["observation"] = d.observation
    return json.dumps(o)


def observation_from_json(s: str) -> TimedObservation:
    o = json.loads(s)
    return TimedObservation(float(o["t"]), None if o.get("observation") is None else float(o["observation"]))


def read_json_observations(path: str) -> List[TimedObservation]:
    with open(path) as f:
        return [observation_from_json(line) for line in f if line.strip()]


def write_json_observations(path: str, data: Iterable[TimedObservation]) -> None:
    with open(path, "w") as f:
        for d in data:
            f.write(observation_to_json(d) + "\n")


# --------------------------------------------------------------------------- state
def state_to_json_obj(state: Sequence[float], leaf_dims: Sequence[int]) -> list:
    """A flat state in Tree.flatten order + the leaf dimensions -> ``[{"value": [..]}, ..]``."""
    out, i = [], 0
    for d in leaf_dims:
        out.append({"value": [float(v) for v in state[i:i + d]]})
        i += d
    return out


def state_from_json_obj(obj: list) -> np.ndarray:
    return np.array([v for leaf in obj for v in (leaf["value"] if isinstance(leaf["value"], list) else [leaf["value"]])],
                    dtype=np.float64)


# --------------------------------------------------------------------------- parameters
def _sde_param_to_obj(p) -> dict:
    if isinstance(p, BrownianParameter):
        return {"m0": p.m0, "c0": p.c0, "sigma": p.sigma}
    if isinstance(p, GenBrownianParameter):
        return {"m0": p.m0, "c0": p.c0, "mu": p.mu, "sigma": p.sigma}
    if isinstance(p, OuParameter):
        return {"m0": p.m0, "c0": p.c0, "phi": p.phi, "mu": p.mu, "sigma": p.sigma}
    raise TypeError(f"{type(p).__name__} has no reference JSON form")


def _vec(v) -> List[float]:
    return [float(x) for x in v] if isinstance(v, list) else [float(v)]   # DenseVector also reads a bare number, :25


def _sde_param_from_obj(o: dict):
    n = len(o)                                                   # the reference dispatches on the field count, :41-45
    if n == 3:
        return BrownianParameter(_vec(o["m0"]), _vec(o["c0"]), _vec(o["sigma"]))
    if n == 4:
        return GenBrownianParameter(_vec(o["m0"]), _vec(o["c0"]), _vec(o["mu"]), _vec(o["sigma"]))
    return OuParameter(_vec(o["m0"]), _vec(o["c0"]), _vec(o["phi"]), _vec(o["mu"]), _vec(o["sigma"]))


def parameters_to_json_obj(p: Parameters) -> list:
    out = []
    for n in p.leaves:
        node = {"sdeParam": _sde_param_to_obj(n.sdeParam)}
        if n.scale is not None:
            node["scale"] = n.scale
        out.append(node)
    return out


def parameters_from_json_obj(obj: list) -> Parameters:
    return Parameters([ParamNode(None if o.get("scale") is None else float(o["scale"]), _sde_param_from_obj(o["sdeParam"]))
                       for o in obj])


def parameters_csv(p: Parameters) -> str:  # Show[Parameters], CsvFormat.scala:41-47
    return ", ".join(repr(float(v)) for v in p.flattenParams())


# --------------------------------------------------------------------------- MCMC output
def metrop_state_to_json(s: MetropState, time: float, leaf_dims: Sequence[int]) -> str:
    state = np.zeros(sum(leaf_dims)) if s.sde is None else s.sde
    return json.dumps({"ll": s.ll, "params": parameters_to_json_obj(s.params),
                       "sde": {"time": time, "state": state_to_json_obj(state, leaf_dims)}, "accepted": s.accepted})


def metrop_state_from_json(line: str) -> MetropState:
    o = json.loads(line)
    return MetropState(float(o["ll"]), parameters_from_json_obj(o["params"]), state_from_json_obj(o["sde"]["state"]),
                       int(o["accepted"]))


def read_pmmh_json(path: str, burn_in: int = 0, thin: int = 1) -> Iterator[MetropState]:
    """Streaming.readPosterior (Streaming.scala:113-140): drop ``burn_in`` lines, keep every ``thin``-th."""
    with open(path) as f:
        for i, line in enumerate(l for l in f if l.strip()):
            if i >= burn_in and (i - burn_in) % thin == 0:
                yield metrop_state_from_json(line)


def metrop_state_csv(s: MetropState) -> str:  # Show[MetropState], CsvFormat.scala:53-56
    return f"{parameters_csv(s.params)}, {s.accepted}"


# --------------------------------------------------------------------------- filter output
def pfout_csv(o: PfOut) -> str:  # Show[PfOut], CsvFormat.scala:75-83
    obs = "NA" if o.observation is None else repr(float(o.observation))
    state = ", ".join(repr(float(v)) for v in o.state)
    ivals = ", ".join(f"{repr(float(c.lower))}, {repr(float(c.upper))}" for c in o.stateIntervals)
    return f"{o.time}, {obs}, {repr(float(o.eta))}, {repr(float(o.etaIntervals.lower))}, {repr(float(o.etaIntervals.upper))}, {state}, {ivals}"


def state_space_csv(s: StateSpace) -> str:  # Show[StateSpace], CsvFormat.scala:49-51
    return f"{s.time}, " + ", ".join(repr(float(v)) for v in s.state)

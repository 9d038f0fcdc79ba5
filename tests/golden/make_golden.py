#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run in the BUILD container).

* densities.json (K1): log-densities from scipy.stats -- the published closed forms of the breeze
  1.0 distributions the reference calls (Poisson.logProbabilityOf, Gaussian.logPdf;
  model/Model.scala:273,230,255).  breeze itself is not under /root/reference and no JVM exists here.
* oracle_runs.json (K5): log-likelihood, ESS trace and ancestor arrays of the CPU restatement
  (oracle/) for the BASELINE configurations at small N under the fixed Philox seed.  These pin the
  oracle against accidental change and give the GPU tests a target that does not depend on the
  oracle being rebuilt on the GPU box.
Nothing here reads /root/reference (it is Scala; there is nothing importable).
"""
import json
import os
import sys

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def densities():
    rng = np.random.default_rng(12345)
    gam = np.concatenate([np.linspace(-6, 6, 25), rng.normal(0, 2, 40)])
    ks = np.array([0, 1, 2, 3, 5, 8, 13, 21, 40, 77, 150, 300, 1000])
    pois = [{"gamma": float(g), "y": float(k) + 0.7 * (i % 2), "logpmf": float(stats.poisson.logpmf(int(k), np.exp(g)))}
            for i, g in enumerate(gam) for k in ks]  # y.toInt truncates: 3.7 -> 3
    gauss = []
    for g in gam[::3]:
        for y in (-3.2, -0.5, 0.0, 0.25, 1.0, 7.5):
            for sd in (0.05, 0.5, 1.0, 3.0):
                gauss.append({"gamma": float(g), "y": y, "sd": sd, "logpdf": float(stats.norm.logpdf(y, loc=g, scale=sd))})
    json.dump({"poisson": pois, "gaussian": gauss, "source": f"scipy {__import__('scipy').__version__}"},
              open(os.path.join(HERE, "densities.json"), "w"))
    print("densities.json:", len(pois), "poisson,", len(gauss), "gaussian")


def oracle_runs():
    import cases
    from oracle import oracle
    runs = {}
    specs = [("c1", cases.c1_model, 1000, 100, 0.0, 0), ("c2", cases.c2_model, 1024, 60, 0.1, 0),
             ("c3", cases.c3_model, 1024, 40, 0.0, 0), ("c4", cases.c4_model, 1024, 12, 0.0, 2),
             ("linear", cases.linear_model, 777, 30, 0.0, 0)]
    for name, mk, n, T, missing, prec in specs:
        model = mk()
        if name == "c4":
            t, y, has = cases.event_times(T)
        elif name == "linear":
            t, y, has = cases.gaussian_series(T)
        else:
            t, y, has = cases.poisson_counts(T, missing=missing)
        o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
        ll, ll_t, ess_t, path = o.filter(t, y, has, want_path=True)
        runs[name] = {"n": n, "T": T, "missing": missing, "lgcp_precision": prec, "ll": ll.hex() if hasattr(ll, "hex") else float(ll).hex(),
                      "ll_t": [float(v).hex() for v in ll_t], "ess_t": [int(v) for v in ess_t],
                      "ancestors_last": [int(a) for a in o.ancestors()],
                      "path": [[float(v).hex() for v in row] for row in path]}
        print(name, "ll =", ll)
    json.dump(runs, open(os.path.join(HERE, "oracle_runs.json"), "w"))


if __name__ == "__main__":
    densities()
    oracle_runs()

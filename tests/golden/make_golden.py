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
    # "next" densities, closed forms of the breeze calls at model/Model.scala:155-160,186-195,298-307,318-336,349-352
    negbin, zipd, bern, stt, beta = [], [], [], [], []
    for g in gam[::4]:
        mu = np.exp(g)
        for k in (0, 1, 3, 10, 47):
            for size in (0.5, 2.5, 30.0):
                negbin.append({"gamma": float(g), "y": float(k), "size": size,
                               "logpmf": float(stats.nbinom.logpmf(k, size, size / (size + mu)))})
            for v in (-2.0, -0.4, 1.3):
                p = 1.0 / (1.0 + np.exp(-v))
                val = np.log(p + (1 - p) * np.exp(-mu)) if k == 0 else np.log(1 - p) + stats.poisson.logpmf(k, mu)
                zipd.append({"gamma": float(g), "y": float(k), "v": v, "logpmf": float(val)})
        if -6 <= g <= 6:
            pr = 1.0 / (1.0 + np.exp(-g))
            bern.append({"gamma": float(g), "y": 1.0, "logp": float(np.log(pr))})
            bern.append({"gamma": float(g), "y": 0.0, "logp": float(np.log1p(-pr))})
        for y in (-2.5, 0.0, 0.8, 4.0):
            for v in (0.3, 1.0, 2.2):
                for df in (1, 5, 30):
                    # the reference multiplies the LOG-pdf by 1/v (model/Model.scala:158), reproduced as written
                    stt.append({"gamma": float(g), "y": y, "v": v, "df": df, "val": float(stats.t.logpdf((y - g) / v, df) / v)})
        for y in (0.02, 0.3, 0.77, 0.999):
            beta.append({"gamma": float(g), "y": y, "logpdf": float(stats.beta.logpdf(y, np.exp(-g), 1.0))})
    json.dump({"poisson": pois, "gaussian": gauss, "negbin": negbin, "zip": zipd, "bernoulli": bern, "student_t": stt, "beta": beta,
               "source": f"scipy {__import__('scipy').__version__}"},
              open(os.path.join(HERE, "densities.json"), "w"))
    print("densities.json:", len(pois), "poisson,", len(gauss), "gaussian")


def oracle_runs():
    import cases
    from oracle import oracle
    runs = {}
    specs = [("c1", cases.c1_model, 1000, 100, 0.0, 0), ("c2", cases.c2_model, 1024, 60, 0.1, 0),
             ("c3", cases.c3_model, 1024, 40, 0.0, 0), ("c4", cases.c4_model, 1024, 12, 0.0, 2),
             ("linear", cases.linear_model, 777, 30, 0.0, 0), ("negbin", cases.negbin_model, 900, 25, 0.0, 0),
             ("zip", cases.zip_model, 640, 25, 0.0, 0), ("bernoulli", cases.bernoulli_model, 512, 25, 0.0, 0),
             ("studentt", cases.studentt_model, 700, 25, 0.0, 0), ("beta", cases.beta_model, 600, 25, 0.0, 0)]
    for name, mk, n, T, missing, prec in specs:
        model = mk()
        if name == "c4":
            t, y, has = cases.event_times(T)
        elif name in ("linear", "studentt"):
            t, y, has = cases.gaussian_series(T)
        elif name == "zip":
            t, y, has = cases.counts_with_zeros(T)
        elif name == "bernoulli":
            t, y, has = cases.binary_series(T)
        elif name == "beta":
            t, y, has = cases.unit_interval_series(T)
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

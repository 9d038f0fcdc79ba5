#!/usr/bin/env python3
"""Single-GPU filters of SEVERAL PROCESSES side by side on this GPU (the way independent PMMH chains or a shared node run them), each against
the oracle: the launches' intra-launch protocols (unit sums -> publisher block, group sums, the fused sums' hold) under the block timing of a
shared GPU.  usage: concurrent_single_soak.py [processes=4] [N=65536] [T=300] [rounds=2]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def worker(idx, n, T, rnd, q):
    import cases
    from composablestatespacemodels_amd.filter import NativePf
    kinds = [("c2_model", 0), ("c1_model", 0), ("c3_model", 0), ("c4_model", 2)]
    name, prec = kinds[(idx + rnd) % len(kinds)]
    model = getattr(cases, name)()
    nn = n + 1024 * idx + 7 * rnd
    if prec:
        t, y, has = cases.event_times(T // 4, horizon=0.1 * (T // 4))
    else:
        t, y, has = cases.poisson_counts(T, seed=100 + idx, missing=0.1)
        y = y.copy(); y[T // 2] = 60.0; has[T // 2] = 1        # an outlying observation: the series holds and redoes it
    g = NativePf(model, nn, cases.SEED + idx, lgcp_precision=prec)
    cut = len(t) // 3
    g.run(t[:cut], y[:cut], has[:cut])
    ll = None
    for lo in range(cut, len(t), 17):
        ll, _, ess_t = g.run_more(t[lo:lo + 17], y[lo:lo + 17], has[lo:lo + 17])
    q.put((idx, name, nn, float(ll), int(ess_t[-1]), g.particles().tobytes(), t, y, has, prec))
    g.close()


def main():
    import torch.multiprocessing as mp
    import cases
    from oracle import oracle
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    ctx = mp.get_context("spawn")
    bad = 0
    for rnd in range(rounds):
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(i, n, T, rnd, q)) for i in range(procs)]
        for p in ps:
            p.start()
        res = [q.get(timeout=600) for _ in ps]
        for p in ps:
            p.join()
        for idx, name, nn, ll, ess, part, t, y, has, prec in sorted(res):
            o = oracle.OraclePf(getattr(cases, name)().descriptor(prec), nn, cases.SEED + idx)
            oll, _, oess, _ = o.filter(t, y, has)
            ok = ll == oll and ess == int(oess[-1]) and part == o.particles().tobytes()
            print(f"round {rnd} process {idx}: {name} N={nn} T={len(t)}: {'identical' if ok else 'DIFFERENT'} (ll {ll!r} vs {oll!r})", flush=True)
            bad += 0 if ok else 1
    print("SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

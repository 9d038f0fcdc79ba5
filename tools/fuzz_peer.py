"""Ad-hoc fuzz of the sharded filter on the peer-written exchange (shards of one process: tests/local_comm.py, LocalCommPeer) against
the single-rank CPU oracle: random world sizes, cloud sizes, models, resamplers, series with missing observations and an outlier, a
continued part.  usage (GPU box): [FUZZ_COMM=peer|collective|trimmed] python tools/fuzz_peer.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import cases
from oracle import oracle
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
from local_comm import LocalComm, LocalCommPeer, LocalCommTrimmed

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(ncases):
    world = int(rng.choice([2, 3, 4, 5, 8]))
    lg = rng.random() < 0.25
    name = "c4_model" if lg else ["c2_model", "c1_model", "c3_model"][int(rng.integers(0, 3))]
    n = int(rng.integers(world * 600, 150_000))
    T = int(rng.integers(5, 12))
    model = getattr(cases, name)()
    if lg:
        t, y, has = cases.event_times(T, horizon=float(rng.uniform(6.0, 14.0)))
    else:
        t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
        y = y.copy()
        if rng.random() < 0.4:
            s = int(rng.integers(1, T)); y[s] = 70.0; has[s] = 1
    prec = 2 if lg else 0
    strat = (not lg) and rng.random() < 0.3
    a = int(rng.integers(2, T - 1))
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec) for r in range(world)]
    if strat:
        for s_ in shards: s_.set_option(2, 1)
    Comm = {"peer": LocalCommPeer, "collective": LocalComm, "trimmed": LocalCommTrimmed}[os.environ.get("FUZZ_COMM", "peer")]
    f = ShardedFilter(shards, Comm(world))
    f.ll_filter(t[:a], y[:a], has[:a], lgcp=lg)
    try:
        ll, ess = f.ll_filter_more(t[a:], y[a:], has[a:], lgcp=lg)
    except RuntimeError as e:          # (a continued series that would need the exact exchange from its start: the whole series in one call instead)
        print(f"case {c}: continued part refused ({str(e)[:60]}...): the series in one call", flush=True)
        a = 0
        ll, ess = f.ll_filter(t, y, has, lgcp=lg)
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED, oracle.RESAMPLE_STRATIFIED) if strat else oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
    oll, _, oess, _ = o.filter(t, y, has)
    part = np.concatenate([s_.particles() for s_ in shards], axis=1)
    ok = (ll == oll) and (ess == oess[-1]) and np.array_equal(part, o.particles())
    print(f"case {c}: {name} world={world} N={n} T={T} split at {a} stratified={strat} peer={f.last_peer} resumes={f.last_resumes} level redos={f.last_level_redos} attempts={f.last_attempts}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    bad += 0 if ok else 1
    for s_ in shards: s_.close()
print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Soak of the peer-written exchange ACROSS PROCESSES sharing this GPU (windows mapped over hipIpc, control plane on gloo): long series on the
merged launch with few eager rows, so that nearly every exchange takes the path behind the headers (rows beyond the eager ones, EXTRA flag),
and every rank's ll / ESS / cloud against the single-rank oracle.  Timing between the processes differs from run to run: what a race in the
flag protocol would need.  usage: ipc_soak.py [world=3] [observations=400] [rounds=3] [eager=8] [outliers | lgcp]
outliers (world 2): two observations outlying enough to rule their reference level out -- the series holds there on every rank and the
observation is redone in place with its level from the all-gathered max (cssm_pf_shard_resume_level).  SOAK_PER_RANK=<particles per rank>
(default 4096); CSSM_PEER_TWO_LAUNCHES=1 / CSSM_GRP_MIN_UNITS=100000 send the series through the staged launches / the launches without
group sums."""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def series(T, outliers):
    import cases as cs
    if outliers == "lgcp":
        return cs.event_times(T, horizon=0.1 * T)
    t, y, has = cs.poisson_counts(T, missing=0.1)
    if outliers == "outliers":
        y = y.copy(); has = has.copy()
        for k in (T // 2 + 7, T - 40):
            y[k] = 40.0; has[k] = 1
    return t, y, has


def rank_main(rank, world, port, n, T, outdir, stratified, outliers):
    import torch
    import torch.distributed as dist
    import cases as cs
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooPeerComm(DistComm):
        peer = os.environ.get("SOAK_PROTOCOL", "peer") == "peer"     # SOAK_PROTOCOL=torch: every exchange through torch.distributed (gloo)

    torch.cuda.set_device(0)
    lgcp = outliers == "lgcp"
    model = cs.c4_model() if lgcp else cs.c2_model()
    t, y, has = series(T, outliers)
    shard = GpuShard(model, n, rank, world, cs.SEED, 0, lgcp_precision=2 if lgcp else 0)
    if stratified:
        shard.set_option(2, 1)
    f = ShardedFilter([shard], GlooPeerComm())
    cut = T // 3
    f.ll_filter(t[:cut], y[:cut], has[:cut], lgcp=lgcp)
    ll, ess = f.ll_filter_more(t[cut:], y[cut:], has[cut:], lgcp=lgcp)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), ll=ll, ess=ess, part=shard.particles(), peer=int(f.last_peer), resumes=int(f.last_resumes), redos=int(f.last_level_redos),
             rows=np.asarray(shard.peer_rows(), dtype=np.int64))
    shard.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    import cases
    from oracle import oracle
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    os.environ["CSSM_PEER_EAGER_ROWS"] = sys.argv[4] if len(sys.argv) > 4 else "8"
    os.environ.setdefault("CSSM_GRP_MIN_UNITS", "1")
    outliers = sys.argv[5] if len(sys.argv) > 5 else ""      # "", "outliers" or "lgcp" (BASELINE configs[3]: event times, sub-stepped OU)
    lgcp = outliers == "lgcp"
    model = cases.c4_model() if lgcp else cases.c2_model()
    t, y, has = series(T, outliers)
    bad = 0
    for rnd in range(rounds):
        for stratified in (False, True):
            n = int(os.environ.get("SOAK_PER_RANK", "4096")) * world + 1024 * rnd
            o = oracle.OraclePf(model.descriptor(2 if lgcp else 0), n, cases.SEED, oracle.RESAMPLE_STRATIFIED if stratified else 0)
            oll, _, oess, _ = o.filter(t, y, has)
            with tempfile.TemporaryDirectory() as d:
                mp.spawn(rank_main, args=(world, 29900 + (os.getpid() + rnd * 2 + int(stratified)) % 90, n, T, d, stratified, outliers), nprocs=world, join=True)
                z = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(world)]
            ok = all(float(q["ll"]) == oll and int(q["ess"]) == int(oess[-1]) and int(q["peer"]) == int(os.environ.get("SOAK_PROTOCOL", "peer") == "peer") for q in z) and \
                np.array_equal(np.concatenate([q["part"] for q in z], axis=1), o.particles())
            if not ok:
                print(f"   oracle ll {oll!r} ess {int(oess[-1])}; ranks: " + "; ".join(f"ll {float(q['ll'])!r} ess {int(q['ess'])} peer {int(q['peer'])}" for q in z) +
                      f"; clouds equal: {np.array_equal(np.concatenate([q['part'] for q in z], axis=1), o.particles())}", flush=True)
            beyond = sum(int(q["rows"][2]) for q in z); seg = sum(int(q["rows"][1]) for q in z)
            print(f"round {rnd} world {world} N {n} T {T} stratified {stratified}: {'identical' if ok else 'DIFFERENT'}; {beyond} of {seg} neighbour segments needed rows beyond "
                  f"the {os.environ['CSSM_PEER_EAGER_ROWS']} eager ones; resumes {[int(q['resumes']) for q in z]} level redos {[int(q['redos']) for q in z]}", flush=True)
            bad += 0 if ok else 1
    print("SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

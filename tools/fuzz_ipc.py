#!/usr/bin/env python3
"""tools/fuzz_peer.py ACROSS PROCESSES: random world sizes (2-5 processes sharing this GPU, windows over hipIpc, control plane on gloo), cloud
sizes, models (BASELINE's C1-C4 and the eight one-component families), resamplers, eager-row counts, launches with and without group sums,
series with missing observations and an outlier, a continued part -- the merged exchange launch under the timing of separate processes --
every rank against the single-rank CPU oracle.  usage (GPU box): python tools/fuzz_ipc.py [cases=12] [seed=1]"""
import os
import sys
import tempfile

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))


def build_case(c, seed):
    import cases
    rng = np.random.default_rng([seed, c])
    world = int(rng.choice([2, 3, 4, 5]))
    lg = rng.random() < 0.25
    name = "c4_model" if lg else ["c2_model", "c1_model", "c3_model", "euler_model"][int(rng.integers(0, 4))]
    n = int(rng.integers(world * 600, 60_000))
    T = int(rng.integers(20, 70))
    if lg:
        t, y, has = cases.event_times(T, horizon=float(rng.uniform(0.08, 0.2)) * T)
    else:
        t, y, has = cases.poisson_counts(T, seed=int(rng.integers(1, 1 << 30)), missing=0.2)
        y = y.copy(); has = has.copy()
        if world == 2 and rng.random() < 0.5:      # (an outlier's offspring reach non-adjacent ranks beyond two: another plan, refused in a continued series)
            s = int(rng.integers(1, T)); y[s] = 70.0; has[s] = 1
    strat = (not lg) and rng.random() < 0.3
    a = int(rng.integers(2, T - 1))
    path = rng.random() < 0.3          # `filter`: the whole series in one call, sampleOne's pick after every observation recorded
    env = {"CSSM_PEER_EAGER_ROWS": str(int(rng.choice([1, 8, 64, 4096]))), "CSSM_GRP_MIN_UNITS": str(int(rng.choice([1, 100000])))}
    return dict(world=world, lg=lg, name=name, n=n, T=T, t=t, y=y, has=has, strat=strat, a=a, env=env, path=path)


def rank_main(rank, world, port, c, seed, outdir):
    import torch
    import torch.distributed as dist
    import cases
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    k = build_case(c, seed)
    os.environ.update(k["env"])
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooPeerComm(DistComm):
        peer = True

    torch.cuda.set_device(0)
    model = getattr(cases, k["name"])()
    shard = GpuShard(model, k["n"], rank, world, cases.SEED, 0, lgcp_precision=2 if k["lg"] else 0)
    if k["strat"]:
        shard.set_option(2, 1)
    f = ShardedFilter([shard], GlooPeerComm())
    t, y, has, a, lg = k["t"], k["y"], k["has"], k["a"], k["lg"]
    whole, path = 0, np.zeros(0)
    if k["path"]:
        whole = 2
        ll, path = f.filter(t, y, has, lgcp=lg)
        ess = -1
    else:
        f.ll_filter(t[:a], y[:a], has[:a], lgcp=lg)
        try:
            ll, ess = f.ll_filter_more(t[a:], y[a:], has[a:], lgcp=lg)
        except RuntimeError:               # (a continued series that would need another plan from its start: the whole series in one call)
            whole = 1
            ll, ess = f.ll_filter(t, y, has, lgcp=lg)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), ll=ll, ess=ess, part=shard.particles(), peer=int(f.last_peer), resumes=int(f.last_resumes),
             redos=int(f.last_level_redos), whole=whole, path=path)
    shard.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    import cases
    from oracle import oracle
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    os.environ.setdefault("CSSM_PEER_TIMEOUT_MS", "8000")
    bad = 0
    for c in range(ncases):
        k = build_case(c, seed)
        model = getattr(cases, k["name"])()
        prec = 2 if k["lg"] else 0
        o = oracle.OraclePf(model.descriptor(prec), k["n"], cases.SEED, oracle.RESAMPLE_STRATIFIED if k["strat"] else 0)
        oll, _, oess, opath = o.filter(k["t"], k["y"], k["has"], want_path=k["path"])
        with tempfile.TemporaryDirectory() as d:
            try:
                mp.spawn(rank_main, args=(k["world"], 29700 + (os.getpid() + c) % 90, c, seed, d), nprocs=k["world"], join=True)
                z = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(k["world"])]
                ok = all(float(q["ll"]) == oll and (k["path"] or int(q["ess"]) == int(oess[-1])) for q in z) and \
                    np.array_equal(np.concatenate([q["part"] for q in z], axis=1), o.particles()) and \
                    (not k["path"] or all(np.array_equal(q["path"], opath) for q in z))
                info = f"peer={[int(q['peer']) for q in z]} resumes={int(z[0]['resumes'])} level redos={int(z[0]['redos'])} whole={int(z[0]['whole'])}"
            except Exception as e:      # noqa: BLE001 -- a rank that failed is the finding
                ok, info = False, f"a rank failed: {str(e)[-300:]}"
        print(f"case {c}: {k['name']} world={k['world']} N={k['n']} T={k['T']} split at {k['a']} stratified={k['strat']} {k['env']} {info}: "
              f"{'identical' if ok else 'DIFFERENT'}", flush=True)
        bad += 0 if ok else 1
    print("FUZZ OK" if bad == 0 else f"FUZZ FAILED: {bad} cases")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

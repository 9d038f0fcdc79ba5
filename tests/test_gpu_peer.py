"""-m gpu: the PEER-WRITTEN exchange of the sharded filter (include/cssm_pf.h, cssm_pf_shard_peer_*): k_boundary_pack writes every
segment straight into the destination rank's receive window and sets a flag; k_offspring_expand_spec polls the flags -- no
collective per observation.  On one GPU the protocol runs (a) between the shards of one process (tests/local_comm.py:
LocalCommPeer -- windows mapped by plain pointers), (b) on a rank's own windows at world 1 under torch.distributed, and (c)
between PROCESSES that share the GPU and map each other's windows through hipIpc handles (control plane over gloo).  Every case
must give the bits of the single-rank CPU oracle -- the bits the collective exchange gives (tests/test_gpu_sharded.py)."""
import os
import sys

import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu


def _oracle_run(model, n, t, y, has, lgcp_precision=0):
    o = oracle.OraclePf(model.descriptor(lgcp_precision), n, cases.SEED)
    ll, ll_t, ess_t, _ = o.filter(t, y, has)
    return ll, ess_t, o.particles()


def _peer_filter(model, n, world, lgcp_precision=0):
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalCommPeer
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=lgcp_precision) for r in range(world)]
    return shards, ShardedFilter(shards, LocalCommPeer(world))


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("name,n", [("c2_model", 6000), ("c1_model", 1000), ("c3_model", 4100)])
def test_peer_exchange_local_shards_match_single_rank_oracle(world, name, n):
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(11, missing=0.2)
    shards, f = _peer_filter(model, n, world)
    ll, ess = f.ll_filter(t, y, has)
    assert f.last_peer and f.last_single and not f.last_from_max and f.last_attempts == 1
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    # a second series on the same windows (the exchange counter and the window alternation go on), then a continued one
    ll2, ess2 = f.ll_filter(t[:7], y[:7], has[:7])
    ll3, ess3 = f.ll_filter_more(t[7:], y[7:], has[7:])
    assert (ll3, ess3) == (oll, oess[-1]) and f.last_peer
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("stratified", [False, True])
@pytest.mark.parametrize("world,n", [(2, 40000), (4, 30000), (8, 100000)])
def test_eager_rows_and_the_needed_rows_beyond_them(world, n, stratified, monkeypatch):
    """The boundary blocks are sized for the worst observation (CAP_SQRT x sqrt(N) rows, at least a unit).  What travels: the `eager` rows
    next to the boundary at once, and of the rows beyond them those the neighbour's slots need -- the pack's row blocks wait for every
    rank's header, count, write them and say how many; the reader waits for that only when the eager rows do not reach (cssm_pf.h:
    cssm_pf_shard_pack_rows_peer).  CSSM_PEER_EAGER_ROWS = 8 sends nearly every observation down the second path, 1 all of them, the
    default (four tiles) hardly any; CSSM_PEER_ALL_ROWS=1 is "every row travels".  All: the bits of the oracle."""
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(14, missing=0.15)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED, oracle.RESAMPLE_STRATIFIED if stratified else 0)
    oll, _, oess, _ = o.filter(t, y, has)
    weighted = int(np.count_nonzero(has))
    for mode in ("8", "1", "default", "all"):
        monkeypatch.delenv("CSSM_PEER_ALL_ROWS", raising=False)
        monkeypatch.delenv("CSSM_PEER_EAGER_ROWS", raising=False)
        if mode == "all":
            monkeypatch.setenv("CSSM_PEER_ALL_ROWS", "1")      # (both are read when a handle is created)
        elif mode != "default":
            monkeypatch.setenv("CSSM_PEER_EAGER_ROWS", mode)
        shards, f = _peer_filter(model, n, world)
        if stratified:
            for s in shards:
                s.set_option(2, 1)
        ll, ess = f.ll_filter(t, y, has)
        assert f.last_peer and f.last_single and f.last_attempts == 1
        assert (ll, ess) == (oll, oess[-1]), mode
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), o.particles())
        rows, cap = [s.peer_rows() for s in shards], f.last_cap
        # ... and a continued series on the same windows
        f.ll_filter(t[:6], y[:6], has[:6])
        assert f.ll_filter_more(t[6:], y[6:], has[6:]) == (oll, oess[-1])
        for s in shards:
            s.close()
        for r, (nrows, nseg, beyond) in enumerate(rows):
            neighbours = (1 if r > 0 else 0) + (1 if r + 1 < world else 0)
            assert nseg == neighbours * weighted, (mode, r, nseg)
            full = nseg * min(cap, n // world)
            if mode == "all":
                assert beyond == 0 and nrows == full, (mode, r, nrows, nseg, beyond)
            elif mode == "default":                                 # four tiles of eager rows: an observation that needs more is the exception
                assert beyond <= nseg // 4 and nseg * min(4096, cap, n // world) <= nrows <= full, (mode, r, nrows, nseg, beyond)
            else:
                assert beyond > 0, (mode, r, beyond, nseg)          # the second path ran: needed rows beyond the 8 (1) eager ones
                assert int(mode) * nseg < nrows < full // 4, (mode, r, nrows, full)   # ... and they are a fraction of the capacity


@pytest.mark.parametrize("world,n", [(2, 3000), (4, 9000), (8, 20000)])
@pytest.mark.parametrize("name", ["c4_model", "lgcp_seasonal_model"])
def test_peer_exchange_lgcp(world, n, name):
    """LGCP (numerics contract v8): the first event's level is the all-gathered max (collective exchange), every later event's
    is predicted and its exchange peer-written."""
    model = getattr(cases, name)()
    t, y, has = cases.event_times(8, horizon=12.0)
    shards, f = _peer_filter(model, n, world, lgcp_precision=2)
    ll, ess = f.ll_filter(t, y, has, lgcp=True)
    assert f.last_peer and not f.last_from_max and f.last_attempts == 1
    oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=2)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_peer_exchange_resumes_capacity_misses_and_redoes_an_outlier_in_place(world):
    """One row per pair: the exchanges miss and are resumed -- each redone through the collective exchange with a larger capacity
    while the kernels enqueued behind it (their packs and polls included) returned at once -- and the series goes on peer-written:
    the exchange counter has moved on by the skipped exchanges, on every rank alike.  An outlying observation holds the series
    the same way: that observation alone is run again with its level from the global max (collective exchange)."""
    model = cases.c2_model()
    n = 3000 * world
    t, y, has = cases.poisson_counts(12, missing=0.1)
    shards, f = _peer_filter(model, n, world)
    f.MIN_CAP, f.CAP_SQRT = 1, 0.0
    ll, ess = f.ll_filter(t, y, has)
    assert f.last_peer and f.last_resumes >= 1 and f.last_attempts == 1
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    y2 = y.copy(); y2[5] = 60.0; has2 = has.copy(); has2[5] = 1
    f.MIN_CAP, f.CAP_SQRT = 1024, 6.0
    ll, ess = f.ll_filter(t, y2, has2)
    # (the outlying observation is run again from the global max, in place: cssm_pf_shard_resume_level -- or the plan moves on to
    #  "exact" where the outlier leaves whole ranks without weight)
    assert f.last_level_redos >= 1 or f.last_attempts >= 2
    oll, oess, opart = _oracle_run(model, n, t, y2, has2)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    # ... and back on the peer windows afterwards
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert f.last_peer and (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


def test_peer_exchange_at_full_size_equals_single_gpu():
    """BASELINE configs[1] at 2^20 particles over 8 shards and configs[3]'s shape (LGCP, 8 x 2^18 here) -- the sizes at which a unit of
    sums has several tiles and a boundary block several units -- equal the single-GPU handle bit for bit."""
    from composablestatespacemodels_amd.filter import NativePf
    for model, n, T, prec, data in ((cases.c2_model(), 1 << 20, 10, 0, cases.poisson_counts), (cases.c4_model(), 1 << 21, 4, 2, cases.event_times)):
        t, y, has = data(T)
        g = NativePf(model, n, cases.SEED, lgcp_precision=prec)
        ll1, _, ess1, _ = g.run(t, y, has)
        ref = g.particles()
        g.close()
        shards, f = _peer_filter(model, n, 8, lgcp_precision=prec)
        ll, ess = f.ll_filter(t, y, has, lgcp=bool(prec))
        assert f.last_peer and (ll, ess) == (ll1, int(ess1[-1]))
        for s in shards:
            np.testing.assert_array_equal(s.particles(), ref[:, s.first:s.first + s.n])
            s.close()


def test_peer_exchange_world1_under_torch_distributed():
    """One shard per process as bench.py --sharded runs it: the library's own loop (cssm_pf_shard_series_peer), the rank's windows
    being its own peer; no RCCL communicator is created for the series."""
    import torch
    import torch.distributed as dist
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for model, n, prec, data in ((cases.c2_model(), 5000, 0, cases.poisson_counts), (cases.c4_model(), 6000, 2, cases.event_times)):
            t, y, has = data(14)
            shard = GpuShard(model, n, 0, 1, cases.SEED, 0, lgcp_precision=prec)
            f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
            oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=prec)
            ll, ess = f.ll_filter(t, y, has, lgcp=bool(prec))
            assert f.last_peer and not f.last_native and (ll, ess) == (oll, oess[-1])
            np.testing.assert_array_equal(shard.particles(), opart)
            f.ll_filter(t[:5], y[:5], has[:5], lgcp=bool(prec))
            shard.profile(True)
            ll, ess = f.ll_filter_more(t[5:], y[5:], has[5:], lgcp=bool(prec))
            prof = shard.profile_read()
            shard.profile(False)
            assert f.last_peer and (ll, ess) == (oll, oess[-1])
            nw = int(np.sum(has[5:])) if not prec else len(t) - 5
            # no collective; the pack blocks lead the grid of the offspring launch (k_exchange_offspring); the first LGCP event of a NEW
            # series goes through the collective exchange -- not in a continued one
            assert prof["collective"][1] == 0 and prof["k_boundary_pack"][1] == 0 and prof["k_offspring_expand_spec"][1] == nw
            shard.close()
            f.comm.close()
    finally:
        dist.destroy_process_group()


def _ipc_rank(rank, world, port, n, T, outdir):
    """One rank of test_peer_exchange_between_processes_sharing_the_gpu (spawned; every rank on cuda:0)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases as cs
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooPeerComm(DistComm):          # control plane on the CPU (gloo), windows on the GPU (IPC)
        peer = True

    torch.cuda.set_device(0)
    model = cs.c2_model()
    t, y, has = cs.poisson_counts(T, missing=0.1)
    shard = GpuShard(model, n, rank, world, cs.SEED, 0)
    f = ShardedFilter([shard], GlooPeerComm())
    ll, ess = f.ll_filter(t, y, has)
    ll2, ess2 = f.ll_filter_more(t[-3:] + T, y[-3:], has[-3:])
    np.savez(os.path.join(outdir, f"r{rank}.npz"), ll=ll, ess=ess, ll2=ll2, ess2=ess2, part=shard.particles(), peer=int(f.last_peer),
             resumes=int(f.last_resumes), stale=int(getattr(f, "peer_probe_stale", -1)), rows=np.asarray(shard.peer_rows(), dtype=np.int64))
    shard.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,grp,mode", [(2, False, ""), (3, False, ""), (2, True, ""), (4, False, ""), (4, True, ""),
                                            (3, True, "stages"), (3, False, "all_rows"), (3, True, "eager8"), (2, False, "eager8"),
                                            (4, True, "stages+eager8")])
def test_peer_exchange_between_processes_sharing_the_gpu(tmp_path, world, grp, mode, monkeypatch):
    """`world` PROCESSES, all on this GPU, each mapping the others' windows through hipIpcOpenMemHandle: the handles, the
    device table, the flags (world x world of them), the self-validating header words and the window alternation as across GPUs (what
    one GPU cannot show is xGMI visibility; the handshake's probe words are read back here too).  Small clouds, so that the processes'
    kernels fit the GPU side by side.  grp: the exchange on the group sums (CSSM_GRP_MIN_UNITS = 1 sends these few units through them).
    World 4 is as far as this goes: a GPU box admits six processes on its card, and the test runner is one of them -- eight ranks
    cannot be started here (the in-process shards of LocalCommPeer cover world 8, without the IPC mapping).
    mode "stages": the library's series loop enqueues the exchange as its three stage launches (headers / needed rows / adopt:
    CSSM_PEER_TWO_LAUNCHES) instead of the merged kernel; "all_rows": every row of the boundary blocks travels (CSSM_PEER_ALL_ROWS);
    "eager8": eight eager rows, so that the rows beyond them -- written behind the headers, waited for by the reader -- carry most exchanges."""
    import torch.multiprocessing as mp
    if grp:
        monkeypatch.setenv("CSSM_GRP_MIN_UNITS", "1")
    if "stages" in mode:
        monkeypatch.setenv("CSSM_PEER_TWO_LAUNCHES", "1")
    if mode == "all_rows":
        monkeypatch.setenv("CSSM_PEER_ALL_ROWS", "1")
    if "eager8" in mode:      # (the default eager count exceeds these shards' capacity: every row would travel at once)
        monkeypatch.setenv("CSSM_PEER_EAGER_ROWS", "8")
    n, T = 4096 * world, 9
    port = 29700 + (os.getpid() % 200) + world
    mp.spawn(_ipc_rank, args=(world, port, n, T, str(tmp_path)), nprocs=world, join=True)
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(T, missing=0.1)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    ll, _, ess_t, _ = o.filter(t, y, has)
    for a, b, h in zip(t[-3:] + T, y[-3:], has[-3:]):
        ll2, ess2 = o.step(a, b, bool(h))
    parts, beyond = [], 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert int(z["peer"]) == 1 and int(z["stale"]) == 0
        assert float(z["ll"]) == ll and int(z["ess"]) == int(ess_t[-1])
        assert float(z["ll2"]) == ll2 and int(z["ess2"]) == ess2
        parts.append(z["part"])
        beyond += int(z["rows"][2])
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), o.particles())
    if "eager8" in mode:
        assert beyond > 0          # some exchange needed rows beyond the eight eager ones: the second path ran across processes


def _soak_rank(rank, world, port, n, T, outdir, lgcp=False):
    """One rank of test_a_capacity_miss_between_processes_holds_every_rank_alike / test_lgcp_between_processes_... (spawned; every rank
    on cuda:0)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases as cs
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooPeerComm(DistComm):
        peer = True

    torch.cuda.set_device(0)
    if lgcp:
        t, y, has = cs.event_times(T, horizon=0.1 * T)
        shard = GpuShard(cs.c4_model(), n, rank, world, cs.SEED, 0, lgcp_precision=2)
    else:
        t, y, has = cs.poisson_counts(T, missing=0.1)
        shard = GpuShard(cs.c2_model(), n, rank, world, cs.SEED, 0)
        shard.set_option(2, 1)                  # stratified: the series that met the capacity miss in tools/ipc_soak.py
    f = ShardedFilter([shard], GlooPeerComm())
    cut = T // 3
    f.ll_filter(t[:cut], y[:cut], has[:cut], lgcp=lgcp)
    ll, ess = f.ll_filter_more(t[cut:], y[cut:], has[cut:], lgcp=lgcp)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), ll=ll, ess=ess, part=shard.particles(), resumes=int(f.last_resumes))
    shard.close()
    dist.barrier()
    dist.destroy_process_group()


def test_a_capacity_miss_between_processes_holds_every_rank_alike(tmp_path, monkeypatch):
    """Three processes on this GPU, a 400-observation series with eight eager rows whose observation 315 needs more rows than the capacity
    holds: every rank's first offspring block raises the hold as soon as it has all headers -- while pack blocks of the SAME launch on
    the same rank may not have stored yet (the blocks of a launch start XCD by XCD; with several processes on the GPU microseconds apart).
    Those blocks must not take the hold for theirs (they did, one run in six: a header or a ticket missing, the neighbour waiting until its
    bound): every rank holds, resumes the observation with a larger capacity, and ends on the oracle's bits.  Repeated: it is a race."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("CSSM_GRP_MIN_UNITS", "1")
    monkeypatch.setenv("CSSM_PEER_EAGER_ROWS", "8")
    monkeypatch.setenv("CSSM_PEER_TIMEOUT_MS", "5000")
    world, n, T = 3, 13312, 400
    t, y, has = cases.poisson_counts(T, missing=0.1)
    o = oracle.OraclePf(cases.c2_model().descriptor(), n, cases.SEED, oracle.RESAMPLE_STRATIFIED)
    oll, _, oess, _ = o.filter(t, y, has)
    for rep in range(4):
        out = tmp_path / f"rep{rep}"
        out.mkdir()
        mp.spawn(_soak_rank, args=(world, 29850 + (os.getpid() + rep) % 40, n, T, str(out)), nprocs=world, join=True)
        z = [np.load(os.path.join(str(out), f"r{r}.npz")) for r in range(world)]
        assert all(float(q["ll"]) == oll and int(q["ess"]) == int(oess[-1]) for q in z)
        assert all(int(q["resumes"]) >= 1 for q in z)      # the series did meet its capacity miss
        np.testing.assert_array_equal(np.concatenate([q["part"] for q in z], axis=1), o.particles())


def test_lgcp_between_processes_every_destination_gets_the_same_header(tmp_path, monkeypatch):
    """Three processes on this GPU, BASELINE configs[3] (LGCP: every event's level is predicted from the max of the event before, which
    travels in the headers), 60 events.  A rank's header blocks -- one per destination -- read the max slots that the first offspring block
    of the SAME launch clears at its end; a header block that ran behind it (the blocks of a launch start XCD by XCD, processes take turns)
    sent its destination a header with another max than everybody else's: levels and likelihoods drifted apart between the ranks (every
    run of tools/ipc_soak.py ... lgcp at world 3 before the clear waited for the header blocks).  All ranks: the oracle's bits."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("CSSM_GRP_MIN_UNITS", "1")
    monkeypatch.setenv("CSSM_PEER_EAGER_ROWS", "8")
    monkeypatch.setenv("CSSM_PEER_TIMEOUT_MS", "5000")
    world, n, T = 3, 12288, 60
    t, y, has = cases.event_times(T, horizon=0.1 * T)
    o = oracle.OraclePf(cases.c4_model().descriptor(2), n, cases.SEED)
    oll, _, oess, _ = o.filter(t, y, has)
    for rep in range(2):
        out = tmp_path / f"rep{rep}"
        out.mkdir()
        mp.spawn(_soak_rank, args=(world, 29810 + (os.getpid() + rep) % 30, n, T, str(out), True), nprocs=world, join=True)
        z = [np.load(os.path.join(str(out), f"r{r}.npz")) for r in range(world)]
        assert [float(q["ll"]) for q in z] == [oll] * world and all(int(q["ess"]) == int(oess[-1]) for q in z)
        np.testing.assert_array_equal(np.concatenate([q["part"] for q in z], axis=1), o.particles())


def test_bench_n_rank_path_rehearsed_with_gpu_shards_sharing_this_gpu():
    """`bench.py --gpus 3 --backend gloo-gpu`: what the driver will run on a node (`bench.py --gpus N` -> torch.distributed.run -> one rank
    per GPU), rehearsed with three ranks that are real GpuShards on THIS GPU: the launcher, the pre-flight walk (windows mapped over
    hipIpc, two handshake rounds with probe words, a 6-observation series on the windows and on the torch.distributed collectives returning
    the same bits on every rank; the library's own RCCL refused with a reason: RCCL cannot put two ranks on one device), the timed legs
    on the peer-written exchange, rank 0's line -- and its ll / ess equal the single-rank oracle's.  Not a measurement."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    K, W, R, per = 5, 8, 2, 20000
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--backend", "gloo-gpu", "--particles", str(per), "--steps", str(K),
                        "--warmup", str(W), "--repeats", str(R), "--launch-timeout", "300", "--stage-timeout", "120"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    ex = j["exchange"]
    walk = {st["protocol"]: st for st in ex["preflight"]}
    assert ex["chosen"] == "peer" and walk["peer"]["ok"] and walk["torch"]["ok"] and not walk["rccl"]["ok"] and walk["rccl"]["why"]
    assert "stale by plain loads: 0" in walk["peer"]["why"]
    assert j["n_gpus"] == 3 and j["config"]["particles_total"] == 3 * per and "receive windows" in j["config"]["workload"]
    assert all(p["legs"][0]["plan"] == "ref" for p in j["per_rank"]) and len(j["per_rank"]) == 3
    sys.path.insert(0, root)
    import bench
    model, t, y, has = bench.build_workload(W + (R + 1) * K, "c2")
    o = oracle.OraclePf(model.descriptor(), 3 * per, 20260101)
    T = W + R * K
    ll, _, ess_t, _ = o.filter(t[:T], y[:T], has[:T])
    assert j["ll"] == ll and j["ess_last"] == int(ess_t[-1])


@pytest.mark.parametrize("grp", [False, True])
def test_a_peer_that_never_delivers_ends_in_an_error_not_a_hang(monkeypatch, grp):
    """Rank 1 maps its windows and then never runs its series: rank 0's exchange kernels wait for its header -- the header flag (small
    shards) or the self-validating header words (group sums) -- for CSSM_PEER_TIMEOUT_MS of WALL-CLOCK time (the constant 100 MHz clock,
    not an iteration count), raise sticky bit 16 and every kernel behind them returns at once; the status read names the cause."""
    import time
    from composablestatespacemodels_amd import _abi
    from composablestatespacemodels_amd.sharded import GpuShard
    monkeypatch.setenv("CSSM_PEER_TIMEOUT_MS", "250")
    if grp:
        monkeypatch.setenv("CSSM_GRP_MIN_UNITS", "1")
    model, n, cap, T = cases.c2_model(), 12000, 1024, 4
    shards = [GpuShard(model, n, r, 2, cases.SEED, 0) for r in range(2)]
    handles = [s.peer_setup(cap) for s in shards]
    for s in shards:
        s.peer_connect(handles)
    t, y, has = cases.poisson_counts(T)
    shards[0].begin(t, y, has)                      # only rank 0 starts its series
    t0 = time.time()
    shards[0].series_peer(0, T, np.ones(T, dtype=np.uint8), cap)
    with pytest.raises(_abi.CssmError, match="did not arrive"):
        shards[0].status(T)
    waited = time.time() - t0
    assert 0.2 < waited < 15.0, waited              # the bound, once (the blocks of the first launch wait side by side; the rest return at once)
    for s in shards:
        s.close()


@pytest.mark.parametrize("peer", [True, False])
@pytest.mark.parametrize("world,n", [(2, 6000), (2, 50000)])
def test_an_outlying_observation_is_redone_in_place_also_in_a_continued_series(world, n, peer):
    """An observation whose reference level its max rules out (sticky bit 4) holds the sharded series at that observation, on every
    rank; cssm_pf_shard_resume_level rewinds to before its propagate and that ONE observation runs again with its level from the
    all-gathered max -- in a series run in one call (no second pass under the "max" plan) and in a CONTINUED one, whose start is
    gone.  On the peer-written exchange only (see ShardedFilter.ll_filter); the bits of the single-rank oracle either way.  (Two
    ranks: a datum outlying enough to rule its level out leaves one particle with all the weight, and with three or more ranks its
    offspring reach non-adjacent ranks -- the exact exchange, tests above.)"""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm, LocalCommPeer
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(14, missing=0.1)
    y = y.copy(); has = has.copy()
    y[9] = 40.0; has[9] = 1                      # outlying: the level (the Poisson mode's log-density) is ruled out by the max
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, (LocalCommPeer if peer else LocalComm)(world))
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    ll, ess = f.ll_filter(t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    if peer:
        assert f.last_level_redos == 1 and f.last_attempts == 1 and f.last_peer
    else:
        # (a collective has overwritten the one receive buffer -- the rows the observation would be propagated from again -- before any
        #  rank knows the verdict: the series is repeated with every level from the max, as ever)
        assert f.last_level_redos == 0 and f.last_attempts == 2 and f.last_from_max
    f.ll_filter(t[:6], y[:6], has[:6])
    if peer:
        ll, ess = f.ll_filter_more(t[6:], y[6:], has[6:])
        assert (ll, ess) == (oll, oess[-1]) and f.last_level_redos == 1
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    else:
        with pytest.raises(RuntimeError):
            f.ll_filter_more(t[6:], y[6:], has[6:])
    for s in shards:
        s.close()

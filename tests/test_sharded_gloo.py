"""World-size-2 (and 3) runs of the sharded-filter orchestration over gloo on CPU.

The orchestration (composablestatespacemodels_amd/sharded.py: stage order, all-reduce / all-gather /
all-to-all-v, split sizes) is the product's; the per-shard compute here is the test-only OracleShard
backend.  The result must equal the single-rank oracle bit for bit -- ll, ess and every particle.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _series(C, T, missing, prec):
    if prec:
        return C.event_times(T)
    t, y, has = C.poisson_counts(T, missing=abs(missing))
    if missing < 0:            # negative `missing`: plant an outlying count, whose reference level the max rules out
        y = y.copy(); y[T // 2] = 60.0; has = has.copy(); has[T // 2] = 1
    return t, y, has


def _worker(rank, world, port, name, n, T, missing, prec, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import cases as C
    from composablestatespacemodels_amd.sharded import DistComm, ShardedFilter
    from oracle_shard import OracleShard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = getattr(C, name)()
        t, y, has = _series(C, T, missing, prec)
        shard = OracleShard(model, n, rank, world, C.SEED, prec)
        f = ShardedFilter([shard], DistComm())
        if missing == 0.05:        # capacity too small for what the exchange needs: the observations that miss are resumed
            f.MIN_CAP, f.CAP_SQRT = 1, 0.0
        ll, ess = f.ll_filter(t, y, has, lgcp=bool(prec), exact=(missing == -0.1))
        sm = f.summary(0.9)     # getIntervals over the shards: eight histogram all-reduces + one of the sums, over gloo
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), **sm)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), ll=ll, ess=ess, part=shard.particles(), redone=shard.redone,
                 attempts=f.last_attempts, cap=-1 if f.last_cap is None else f.last_cap, single=int(f.last_single), resumes=int(f.last_resumes), from_max=int(f.last_from_max))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,n,T,missing,prec", [
    (2, "c2_model", 301, 7, 0.2, 0),
    (3, "c1_model", 200, 6, 0.0, 0),
    (2, "c4_model", 150, 4, 0.0, 2),       # LGCP: the level is the global max (all-gather + sums before the all-to-all)
    (3, "c4_model", 200, 5, 0.0, 1),
    (2, "c2_model", 256, 6, -0.1, 0),      # an outlying observation inside a series forced onto the exact exchange
    (2, "c2_model", 256, 12, -0.2, 0),     # ... inside a single-collective series: voided, repeated with levels from the max
    (3, "c2_model", 300, 12, 0.05, 0),     # capacity misses: resumed in place
    (2, "c3_model", 200, 14, 0.0, 0),
])
def test_sharded_orchestration_over_gloo_matches_single_rank(tmp_path, world, name, n, T, missing, prec):
    """ONE all-to-all per observation carries sums and boundary particles (ordinary series); all-gather + host-read sizes +
    all-to-all-v (LGCP, forced, or the repetition after an outlying observation)."""
    port = 29600 + (os.getpid() % 300) + world
    mp.spawn(_worker, args=(world, port, name, n, T, missing, prec, str(tmp_path)), nprocs=world, join=True)
    model = getattr(cases, name)()
    t, y, has = _series(cases, T, missing, prec)
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
    ll, _, ess_t, _ = o.filter(t, y, has)
    parts = []
    om, olo, ohi, oem, oel, oeu = o.summary(0.9)
    for r in range(world):
        sm = np.load(os.path.join(str(tmp_path), f"s{r}.npz"))
        np.testing.assert_array_equal(sm["state_lower"], olo)
        np.testing.assert_array_equal(sm["state_upper"], ohi)
        assert float(sm["eta_lower"]) == oel and float(sm["eta_upper"]) == oeu
        np.testing.assert_allclose(sm["state_mean"], om, rtol=1e-12)
        np.testing.assert_allclose(float(sm["eta_of_mean"]), oem, rtol=1e-12)
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        # sums are a pass of their own (cssm_pf_shard_sums) wherever the level comes from the global max: the FIRST event of an
        # LGCP series (contract v8: every later one predicts its level from the event before), every observation of the
        # repetition after an outlying observation and of a series on the exact exchange; else none
        weighted = int(np.sum(has)) if not prec else T
        assert int(z["redone"]) == (1 if prec else (weighted if missing in (-0.2, -0.1) else 0))
        resumed = missing == 0.05                       # capacity misses are resumed, not repeated from the start
        assert int(z["attempts"]) == (2 if missing == -0.2 else 1)
        assert (int(z["resumes"]) >= 1) == resumed
        assert int(z["single"]) == (0 if missing == -0.1 else 1)
        assert int(z["from_max"]) == (1 if missing == -0.2 else 0)      # (an LGCP series runs the "ref" plan)
        assert float(z["ll"]) == ll and int(z["ess"]) == int(ess_t[-1])
        parts.append(z["part"])
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), o.particles())


def test_local_comm_matches_single_rank_on_cpu():
    """The single-process multi-shard emulation (LocalComm) used by the GPU tests, here with the oracle backend."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    from local_comm import LocalComm
    from oracle_shard import OracleShard
    model = cases.c2_model()
    n, world = 257, 4
    t, y, has = cases.poisson_counts(6, missing=0.2)
    shards = [OracleShard(model, n, r, world, cases.SEED) for r in range(world)]
    ll, ess = ShardedFilter(shards, LocalComm(world)).ll_filter(t, y, has)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    oll, _, oess, _ = o.filter(t, y, has)
    assert (ll, ess) == (oll, int(oess[-1]))
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), o.particles())
    sm, (om, olo, ohi, oem, oel, oeu) = ShardedFilter(shards, LocalComm(world)).summary(0.975), o.summary(0.975)
    np.testing.assert_array_equal(sm["state_lower"], olo)
    np.testing.assert_array_equal(sm["state_upper"], ohi)
    assert (sm["eta_lower"], sm["eta_upper"]) == (oel, oeu)
    np.testing.assert_allclose(sm["state_mean"], om, rtol=1e-12)


def _strat_worker(rank, world, port, n, T, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import cases as C
    from composablestatespacemodels_amd.sharded import DistComm, ShardedFilter
    from oracle_shard import OracleShard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, y, has = C.poisson_counts(T, missing=0.1)
        shard = OracleShard(C.c2_model(), n, rank, world, C.SEED, 0, resampler=1)
        ll, ess = ShardedFilter([shard], DistComm()).ll_filter(t, y, has)
        np.savez(os.path.join(out_dir, f"q{rank}.npz"), ll=ll, ess=ess, part=shard.particles())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stratified_resampling_matches_single_rank(tmp_path, world):
    """Resampling.stratifiedResampling (model/Resampling.scala:78-86) over shards, orchestration rehearsed on the CPU: in one process
    (LocalComm, 4 shards) and over gloo (`world` processes) -- the single-rank oracle's bits with the same resampler."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    from local_comm import LocalComm
    from oracle_shard import OracleShard
    model = cases.c2_model()
    n, T = 300, 7
    t, y, has = cases.poisson_counts(T, missing=0.1)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED, oracle.RESAMPLE_STRATIFIED)
    oll, _, oess, _ = o.filter(t, y, has)
    shards = [OracleShard(model, n, r, 4, cases.SEED, 0, resampler=1) for r in range(4)]
    assert ShardedFilter(shards, LocalComm(4)).ll_filter(t, y, has) == (oll, int(oess[-1]))
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), o.particles())
    port = 29900 + (os.getpid() % 90) + world
    mp.spawn(_strat_worker, args=(world, port, n, T, str(tmp_path)), nprocs=world, join=True)
    parts = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"q{r}.npz"))
        assert float(z["ll"]) == oll and int(z["ess"]) == int(oess[-1])
        parts.append(z["part"])
    np.testing.assert_array_equal(np.concatenate(parts, axis=1), o.particles())


@pytest.mark.parametrize("which", ["c2", "c4"])
def test_bench_launches_its_own_ranks_over_gloo(which):
    """`python bench.py --gpus 2` without a launcher starts its own two ranks (children, before any GPU call in the
    parent), and hands rank 0's JSON line on: rehearsed here over gloo with the oracle shard backend.  The warm-up
    observations start the sharded filter, every timed leg continues it (the sharded ll_filter_more); c2 = BASELINE
    configs[1] weak-scaled, c4 = configs[3] (LGCP, every level from the global max) strong-scaled."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    K, W, R = 5, 2, 2
    n_arg = 300 if which == "c2" else 400           # c2: per GPU; c4: in total
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W),
                        "--repeats", str(R), "--particles", str(n_arg), "--model", which, "--backend", "gloo", "--launch-timeout", "240"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    n_total = 600 if which == "c2" else 400
    assert j["n_gpus"] == 2 and j["steps"] == K and j["repeats"] == R and len(j["wall_ms_each"]) == R
    assert j["config"]["particles_total"] == n_total and j["config"]["particles_per_gpu"] == n_total // 2
    assert j["scaling"] == ("weak" if which == "c2" else "strong")
    assert j["exchange"]["backend"] == "gloo"
    # the pre-flight walk: peer-written windows and the library's own RCCL are refused on CPU ranks, with a reason each, WITHOUT aborting;
    # the legs run on the torch.distributed collectives, whose short series every rank agreed on bit for bit
    walk = j["exchange"]["preflight"]
    assert [st["protocol"] for st in walk] == ["peer", "rccl", "torch"] and [st["ok"] for st in walk] == [False, False, True]
    assert all(st["why"] for st in walk) and j["exchange"]["chosen"] == "torch" and len(j["exchange"]["fallbacks"]) == 2
    assert "torch.distributed" in j["config"]["workload"]
    assert [p["rank"] for p in j["per_rank"]] == [0, 1]
    assert all(len(p["legs"]) == R and p["legs"][0]["plan"] == "ref" for p in j["per_rank"])
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    model, t, y, has = bench.build_workload(W + (R + 1) * K, which)
    o = oracle.OraclePf(model.descriptor(2 if which == "c4" else 0), n_total, 20260101)
    T = W + R * K                                    # the line reports the filter after its last timed leg
    ll, _, ess_t, _ = o.filter(t[:T], y[:T], has[:T])
    assert j["ll"] == ll and j["ess_last"] == int(ess_t[-1])


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies makes the launcher exit non-zero instead of hanging (here: an impossible particle count)."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--particles", "-1", "--backend", "gloo", "--launch-timeout", "120"],
                       capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_rank_that_hangs_in_a_stage_leaves_with_a_non_zero_status():
    """--stage-timeout: a stage of a rank (here the first pre-flight series, with a deadline no series can meet) that does not end makes
    the rank say where it hung and leave with a non-zero status; the launcher ends the other rank and reports failure -- no line, no hang."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--particles", "200000", "--backend", "gloo", "--launch-timeout", "120", "--stage-timeout", "0.05"],
                       capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode != 0
    assert "did not finish within" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_continued_sharded_series_equals_the_whole_series_on_cpu():
    """ll_filter(t[:a]) + ll_filter_more(t[a:]) = ll_filter(t): the sharded cssm_pf_ll_filter_more, here through the CPU stand-in
    (LocalComm; capacity misses resumed inside the continued part) -- tests/test_gpu_sharded.py runs the same on the GPU."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    from local_comm import LocalComm
    from oracle_shard import OracleShard
    model = cases.c2_model()
    n, world, T, a = 300, 3, 12, 5
    t, y, has = cases.poisson_counts(T, missing=0.1)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    oll, _, oess, _ = o.filter(t, y, has)
    for tiny in (False, True):
        shards = [OracleShard(model, n, r, world, cases.SEED) for r in range(world)]
        f = ShardedFilter(shards, LocalComm(world))
        if tiny:
            f.MIN_CAP, f.CAP_SQRT = 1, 0.0
        f.ll_filter(t[:a], y[:a], has[:a])
        ll, ess = f.ll_filter_more(t[a:], y[a:], has[a:])
        assert (ll, ess) == (oll, int(oess[-1]))
        assert (f.last_resumes >= 1) == tiny
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), o.particles())


def test_sharded_filter_returns_the_sampled_path_on_cpu():
    """ShardedFilter.filter = `filter` of ParticleFilter.scala:152-158 over shards: ll and the path of uniformly picked particles
    (row 0: of the initial cloud; the rank that owns the picked global slot records it) equal the single-rank oracle's."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    from local_comm import LocalComm
    from oracle_shard import OracleShard
    model = cases.c2_model()
    n, world, T = 300, 3, 9
    t, y, has = cases.poisson_counts(T, missing=0.2)
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED)
    oll, _, _, opath = o.filter(t, y, has, want_path=True)
    shards = [OracleShard(model, n, r, world, cases.SEED) for r in range(world)]
    ll, path = ShardedFilter(shards, LocalComm(world)).filter(t, y, has)
    assert ll == oll
    np.testing.assert_array_equal(path, opath)

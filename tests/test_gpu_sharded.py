"""-m gpu: the sharded (multi-GPU) stage kernels and orchestration, checked bit for bit against the
single-rank oracle.

One GPU is available, so world > 1 is exercised with all shards in this process and the collectives
replaced by tensor copies (LocalComm); the RCCL code path itself is exercised at world = 1
(DistComm over backend "nccl").  The gloo world-size-2 test of the same orchestration is in
tests/test_sharded_gloo.py (CPU).
"""
import os

import numpy as np
import pytest

import cases
from oracle import oracle

pytestmark = pytest.mark.gpu


def _oracle_run(model, n, t, y, has, lgcp_precision=0):
    o = oracle.OraclePf(model.descriptor(lgcp_precision), n, cases.SEED)
    ll, ll_t, ess_t, _ = o.filter(t, y, has)
    return ll, ess_t, o.particles()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("name,n", [("c2_model", 6000), ("c1_model", 1000), ("c3_model", 4100)])
def test_local_shards_match_single_rank_oracle(world, name, n):
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(9, missing=0.2)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert ll == oll
    assert ess == oess[-1]
    got = np.concatenate([s.particles() for s in shards], axis=1)
    np.testing.assert_array_equal(got, opart)
    for s in shards:
        assert s.result() == (ll, ess)
        s.close()


def test_local_shards_degenerate_weights():
    """An informative first observation concentrates the weight on few particles: candidates for a
    rank then come from other ranks only."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.linear_model(obs_sd=0.001)
    n, world = 4096, 4
    t = np.array([0.0, 1.0, 2.0]); y = np.array([3.0, 3.1, 2.9]); has = np.ones(3, dtype=np.uint8)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)


@pytest.mark.parametrize("world,n", [(2, 3000), (4, 9000), (8, 20000)])
@pytest.mark.parametrize("name", ["c4_model", "lgcp_seasonal_model"])
def test_local_shards_lgcp(world, n, name):
    """LGCP series (BASELINE config 4 is the 8-GPU one).  Numerics contract v8: the level of an event is predicted from the max
    of the event before, so the series runs the "ref" plan -- ONE all-to-all per event, the sums formed inside the propagate --
    except its first event (nothing precedes it: all-gather of the maxima + sums relative to the max ahead of the all-to-all);
    nothing read by the host; bit-identical to the single-rank oracle for 2, 4 and 8 shards.  Also through the exact exchange."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    t, y, has = cases.event_times(7, horizon=12.0)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=2) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=2)
    for exact in (False, True):
        ll, ess = f.ll_filter(t, y, has, lgcp=True, exact=exact)
        assert f.last_single == (not exact) and not f.last_from_max and f.last_attempts == 1
        assert (ll, ess) == (oll, oess[-1])
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world", [2, 8])
def test_full_size_local_shards_equal_single_gpu(world):
    """BASELINE config 2 at its full size (N = 2^20), split into 2 and 8 local shards: ll, ess and the particles equal the
    single-GPU filter of the same N bit for bit (the shard count is not allowed to show in any result)."""
    from composablestatespacemodels_amd.filter import NativePf
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c2_model()
    n, T = 1 << 20, 24
    t, y, has = cases.poisson_counts(T)
    g = NativePf(model, n, cases.SEED)
    ll1, _, ess1, _ = g.run(t, y, has)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has)
    assert (ll, ess) == (ll1, int(ess1[-1]))
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), g.particles())
    for s in shards:
        s.close()
    g.close()


def test_full_size_lgcp_in_its_sharded_shape_equals_single_gpu():
    """BASELINE configs[3] as BASELINE shapes it -- a log-Gaussian Cox process, N = 2^24 particles over 8 shards of 2^21
    (FilterLgcp, ParticleFilter.scala:184-226; precision 2) -- on one GPU with the collectives as tensor copies: the levels are
    predicted (plan "ref": one all-to-all per event; the first event's level is the all-gathered max), and ll, ess and every
    particle equal the single-GPU handle of the same N bit for bit."""
    from composablestatespacemodels_amd.filter import NativePf
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c4_model()
    n, world, T = 1 << 24, 8, 3
    t, y, has = cases.event_times(T)
    g = NativePf(model, n, cases.SEED, lgcp_precision=2)
    ll1, _, ess1, _ = g.run(t, y, has)
    ref = g.particles()
    g.close()
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=2) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has, lgcp=True)
    assert f.last_single and not f.last_from_max and f.last_attempts == 1
    assert (ll, ess) == (ll1, int(ess1[-1]))
    for r, s in enumerate(shards):
        lo, m = s.first, s.n
        np.testing.assert_array_equal(s.particles(), ref[:, lo:lo + m], err_msg=f"shard {r}")
        s.close()


def test_trimmed_exchange_lgcp_world8():
    """The same plan through the trimmed exchange (whole segments between adjacent ranks, 12 header words otherwise; the rest
    of every receive segment NaN): 8 shards, against the oracle."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalCommTrimmed
    model = cases.c4_model()
    n, world = 24000, 8
    t, y, has = cases.event_times(7, horizon=12.0)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=2) for r in range(world)]
    f = ShardedFilter(shards, LocalCommTrimmed(world))
    ll, ess = f.ll_filter(t, y, has, lgcp=True)
    assert f.last_single and not f.last_from_max and f.last_attempts == 1
    oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=2)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("name,prec,n,world", [("c2_model", 0, 9000, 4), ("c3_model", 0, 5000, 2), ("c4_model", 2, 6000, 3)])
def test_continued_sharded_series_equals_the_whole_series(name, prec, n, world):
    """ll_filter(t[:a]) + ll_filter_more(t[a:]) leaves the bits of ll_filter(t): the sharded cssm_pf_ll_filter_more
    (cssm_pf_shard_continue: no new cloud, the clock and the observation count go on; Flow.scan handed the next elements,
    ParticleFilter.scala:163-166) -- with the ordinary capacity and with one row per pair (misses resumed inside the continued
    part, whose records are addressed from 0 again)."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    T, a = 13, 5
    t, y, has = cases.event_times(T, horizon=12.0) if prec else cases.poisson_counts(T, missing=0.1)
    oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=prec)
    for tiny in (False, True):
        shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec) for r in range(world)]
        f = ShardedFilter(shards, LocalComm(world))
        if tiny:
            f.MIN_CAP, f.CAP_SQRT = 1, 0.0
        f.ll_filter(t[:a], y[:a], has[:a], lgcp=bool(prec))
        ll, ess = f.ll_filter_more(t[a:a + 4], y[a:a + 4], has[a:a + 4], lgcp=bool(prec))
        ll, ess = f.ll_filter_more(t[a + 4:], y[a + 4:], has[a + 4:], lgcp=bool(prec))
        assert (ll, ess) == (oll, oess[-1]), (tiny, f.last_resumes)
        assert f.last_attempts == 1
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
        for s in shards:
            s.close()


@pytest.mark.parametrize("name,prec,n,world,tiny", [("c2_model", 0, 9000, 4, False), ("c2_model", 0, 9000, 4, True), ("c3_model", 0, 5000, 2, False),
                                                    ("c4_model", 2, 6000, 3, False)])
def test_sharded_filter_returns_the_sampled_path(name, prec, n, world, tiny):
    """ShardedFilter.filter = `filter` (ParticleFilter.scala:152-158) over shards -- what a BootstrapFilter over several GPUs hands
    PMMH (package.scala:24): ll and one uniformly picked particle of the initial cloud and of the cloud after every observation
    (Resampling.sampleOne on the GLOBAL slot index; the owning rank records it, also when the slot's ancestor is a row received
    from a neighbour, also through resumed capacity misses) equal the single-rank oracle's path bit for bit."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    T = 11
    t, y, has = cases.event_times(T, horizon=12.0) if prec else cases.poisson_counts(T, missing=0.2)
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
    oll, _, _, opath = o.filter(t, y, has, want_path=True)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    if tiny:
        f.MIN_CAP, f.CAP_SQRT = 1, 0.0
    ll, path = f.filter(t, y, has, lgcp=bool(prec))
    assert ll == oll and (f.last_resumes >= 1) == tiny
    np.testing.assert_array_equal(path, opath)
    for s in shards:
        s.close()


def _assert_summary_matches(got, want):
    m, lo, hi, em, el, eu = want
    np.testing.assert_array_equal(got["state_lower"], lo)        # order statistics: exact
    np.testing.assert_array_equal(got["state_upper"], hi)
    assert got["eta_lower"] == el and got["eta_upper"] == eu
    np.testing.assert_allclose(got["state_mean"], m, rtol=1e-12, atol=0)   # fp64 sums in another order
    np.testing.assert_allclose(got["eta_of_mean"], em, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name,prec,n,world,interval", [("c2_model", 0, 9001, 4, 0.975), ("c2_model", 0, 9001, 2, 0.5), ("c1_model", 0, 3000, 3, 0.99),
                                                         ("c3_model", 0, 5000, 2, 0.975), ("gen_brownian_seasonal_gaussian", 0, 4100, 8, 0.9), ("c4_model", 2, 6000, 3, 0.975)])
def test_sharded_summary_equals_the_single_rank_summary(name, prec, n, world, interval):
    """ShardedFilter.summary = getIntervals (ParticleFilter.scala:415-424) of the sharded cloud: credible-interval order statistics
    of every latent component and of eta are GLOBAL ranks (one all-reduce of the byte histograms per radix pass), equal to the
    oracle's over the single-rank cloud bit for bit; the means agree to the order of summation."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    T = 9
    t, y, has = cases.event_times(T, horizon=12.0) if prec else cases.poisson_counts(T, missing=0.2)
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED)
    oll = o.filter(t, y, has)[0]
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, _ = f.ll_filter(t, y, has, lgcp=bool(prec))
    assert ll == oll
    _assert_summary_matches(f.summary(interval), o.summary(interval))
    # a second summary of the same cloud (buffers reused), and one after more observations
    _assert_summary_matches(f.summary(0.6), o.summary(0.6))
    t2 = t[-1] + np.arange(1, 4) * 0.7
    y2 = np.ascontiguousarray(y[-3:][::-1])   # (values of the series' own scale: a continued series cannot change plan)
    if prec:
        f.ll_filter_more(t2, y2, lgcp=True)
    else:
        f.ll_filter_more(t2, y2)
    for a, b in zip(t2, y2):
        o.step(a, b)
    _assert_summary_matches(f.summary(interval), o.summary(interval))
    for s in shards:
        s.close()


def test_rccl_world1_matches_oracle(monkeypatch):
    """The COLLECTIVE exchange over RCCL (CSSM_SHARD_PEER=0: the peer-written exchange is the default where the communicator
    offers it, tests/test_gpu_peer.py), the library issuing the collectives itself."""
    import torch
    import torch.distributed as dist
    from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
    monkeypatch.setenv("CSSM_SHARD_PEER", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model = cases.c2_model()
        n = 5000
        t, y, has = cases.poisson_counts(15)
        shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
        f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
        oll, oess, opart = _oracle_run(model, n, t, y, has)
        for exact in (False, True):          # single-collective exchange (equal-split all-to-all) and exact exchange over RCCL
            ll, ess = f.ll_filter(t, y, has, exact=exact)
            assert (ll, ess) == (oll, oess[-1])
            np.testing.assert_array_equal(shard.particles(), opart)
            if not exact:                    # the library drove the collectives of the series itself (cssm_pf_shard_series_rccl)
                assert f.last_native, "the native RCCL series loop was not taken"
                assert f.last_single
        # the library's loop continues a running filter (cssm_pf_shard_continue) with the same bits, and its event brackets see
        # every kernel and the collective of every weighted observation
        f.ll_filter(t[:6], y[:6], has[:6])
        shard.profile(True)
        ll, ess = f.ll_filter_more(t[6:], y[6:], has[6:])
        prof = shard.profile_read()
        shard.profile(False)
        assert f.last_native and (ll, ess) == (oll, oess[-1])
        np.testing.assert_array_equal(shard.particles(), opart)
        nw = int(np.sum(has[6:]))
        assert prof["k_propagate"][1] == len(t) - 6 and prof["collective"][1] == nw
        assert prof["k_boundary_pack"][1] == nw and prof["k_offspring_expand_spec"][1] == nw
        # `filter` through the library's loop: the path rows are recorded on the stream, combined with one all-reduce at the end
        ll, path = f.filter(t, y, has)
        opath = oracle.OraclePf(model.descriptor(), n, cases.SEED).filter(t, y, has, want_path=True)[3]
        assert f.last_native and ll == oll
        np.testing.assert_array_equal(path, opath)
        # getIntervals of the sharded cloud: the histogram all-reduces run over RCCL
        osum = oracle.OraclePf(model.descriptor(), n, cases.SEED)
        osum.filter(t, y, has)
        _assert_summary_matches(f.summary(0.975), osum.summary(0.975))
        # ... with the all-to-all-v of the library's trimmed exchange (mode 3 forces it at any world size; at world 1 it
        # carries the rank's own header)
        f4 = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
        f4.SINGLE_MODE = 3
        ll, ess = f4.ll_filter(t, y, has)
        assert f4.last_native and f4.last_single and (ll, ess) == (oll, oess[-1])
        np.testing.assert_array_equal(shard.particles(), opart)
        f4.comm.close()
        # an LGCP series with the library's own loop: the first event with an all-gather of the maxima ahead of its all-to-all (mode 1 + 4),
        # every later one on its predicted level (mode 1)
        lm = cases.c4_model()
        lt, ly, lh = cases.event_times(9)
        lshard = GpuShard(lm, 6000, 0, 1, cases.SEED, 0, lgcp_precision=2)
        fl = ShardedFilter([lshard], DistComm(device=torch.device("cuda", 0)))
        ll, ess = fl.ll_filter(lt, ly, lh, lgcp=True)
        lo = _oracle_run(lm, 6000, lt, ly, lh, lgcp_precision=2)
        assert fl.last_native and fl.last_single and not fl.last_from_max and (ll, ess) == (lo[0], lo[1][-1])
        np.testing.assert_array_equal(lshard.particles(), lo[2])
        lshard.close()
        fl.comm.close()
        # the same series with the collectives issued through torch.distributed: identical bits
        os.environ["CSSM_SHARD_NATIVE"] = "0"
        try:
            f2 = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
            ll, ess = f2.ll_filter(t, y, has)
            assert not f2.last_native and (ll, ess) == (oll, oess[-1])
            np.testing.assert_array_equal(shard.particles(), opart)
        finally:
            del os.environ["CSSM_SHARD_NATIVE"]
        shard.close()
        f.comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_local_shards_outlying_observation_second_attempt(world):
    """The gathered max rules the reference level out at one step: shard_sums + a second all-gather, same bits."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c2_model()
    n = 5000
    t, y, has = cases.poisson_counts(8)
    y = y.copy(); y[3] = 60.0
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,n,T", [("c2_model", 20000, 30), ("c3_model", 9000, 16)])
def test_single_collective_series_matches_oracle(world, name, n, T):
    """ll_filter: the host-read-free exchange -- ONE all-to-all per observation carrying sums and boundary particles."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(T, missing=0.1)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    ll, ess = f.ll_filter(t, y, has)
    assert f.last_attempts == 1 and f.last_cap is not None and f.last_single
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    # the same handles run another series (records, capacity buffers and sticky bits start afresh)
    t2, y2, has2 = cases.poisson_counts(12, seed=7)
    ll2, ess2 = f.ll_filter(t2, y2, has2)
    oll2, oess2, _ = _oracle_run(model, n, t2, y2, has2)
    assert (ll2, ess2) == (oll2, oess2[-1])
    for s in shards:
        s.close()


def _mirror_plan(model, n, world, t, y, has, lgcp=False, prec=0, min_cap=None, cap_sqrt=None):
    """What the orchestration does with this series -- (attempts, resumes, single, from_max) -- found by running the SAME
    ShardedFilter over the CPU stand-in (tests/oracle_shard.py; every verdict there is a function of the same bits): the GPU
    run must take exactly that path, not merely arrive at the right numbers by some path."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    from local_comm import LocalComm
    from oracle_shard import OracleShard
    shards = [OracleShard(model, n, r, world, cases.SEED, prec) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    if min_cap is not None:
        f.MIN_CAP, f.CAP_SQRT = min_cap, cap_sqrt
    ll, ess = f.ll_filter(t, y, has, lgcp=lgcp)
    return (f.last_attempts, f.last_resumes, f.last_single, f.last_from_max), (ll, ess)


@pytest.mark.parametrize("why", ["capacity", "outlier", "outlier+capacity"])
def test_series_resumes_a_capacity_miss_and_repeats_after_an_outlying_observation(why):
    """capacity: one row per pair cannot hold the exchange (sticky bit 8): the observations that miss are resumed in place.
    outlier: the max rules a reference level out (bit 4): the series is repeated with every level from the global max.
    outlier+capacity: both -- the repetition (plan "max": all-gather + shard_sums before the all-to-all) ALSO misses its
    capacity and is resumed: the level, the unit sums and the exported words of the observation the series holds at must
    survive the observations enqueued behind it (round 2 lost them: NaN levels for LGCP, a spurious third attempt here)."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c2_model()
    # (a count of 60 also concentrates the weight on very few particles: at world 2 every rank is adjacent to the other, so
    #  boundary blocks -- whole shards if need be -- always cover; at world 4 such a step ends on the exact exchange)
    n, world, T = 12000, (4 if why == "capacity" else 2), 14
    t, y, has = cases.poisson_counts(T)
    if why != "capacity":
        y = y.copy(); y[9] = 60.0
    tiny = why != "outlier"
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    if tiny:
        f.MIN_CAP, f.CAP_SQRT = 1, 0.0
    ll, ess = f.ll_filter(t, y, has)
    plan, _ = _mirror_plan(model, n, world, t, y, has, min_cap=1 if tiny else None, cap_sqrt=0.0 if tiny else None)
    assert (f.last_attempts, f.last_resumes, f.last_single, f.last_from_max) == plan
    if why == "capacity":
        assert plan[0] == 1 and plan[1] >= 1
    elif why == "outlier":
        assert plan[0] == 2 and plan[2] and plan[3]
    else:
        assert plan[0] == 2 and plan[1] >= 1 and plan[2] and plan[3]      # resumed INSIDE the "max" plan
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world,n", [(2, 3000), (4, 9000)])
def test_lgcp_series_resumes_a_capacity_miss(world, n):
    """An LGCP series (plan "ref" on predicted levels, its first event on the all-gathered max): with one row per pair its
    exchanges miss and are resumed in place: ONE attempt, at least one resume, the oracle's bits."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c4_model()
    t, y, has = cases.event_times(8, horizon=12.0)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=2) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    f.MIN_CAP, f.CAP_SQRT = 1, 0.0
    ll, ess = f.ll_filter(t, y, has, lgcp=True)
    plan, _ = _mirror_plan(model, n, world, t, y, has, lgcp=True, prec=2, min_cap=1, cap_sqrt=0.0)
    assert (f.last_attempts, f.last_resumes, f.last_single, f.last_from_max) == plan
    assert plan[0] == 1 and plan[1] >= 1 and plan[2] and not plan[3]
    oll, oess, opart = _oracle_run(model, n, t, y, has, lgcp_precision=2)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("d,n,T", [(1, 3000, 12), (2, 4099, 12), (4, 5001, 12), (6, 4100, 12), (8, 3073, 12), (16, 4097, 12)])
def test_local_shards_every_staging_layout_with_remote_candidates(world, d, n, T):
    """k_propagate stages the next tile through LDS with 16 bytes per element (IT * d <= 9) or 8 bytes (two dword
    fetches); in the sharded filter part of the gathered states are candidates received from other ranks (second
    source, rows of d + 1 doubles in the single-collective series).  Odd shard starts also exercise the unpaired
    normal streams.  Both exchanges against the oracle."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.dim_model(d)
    t, y, has = cases.poisson_counts(T, missing=0.15)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    for exact in (False, True):
        ll, ess = f.ll_filter(t, y, has, exact=exact)
        assert (ll, ess) == (oll, oess[-1])
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("n,world", [(8 * 4096, 2), (8 * 4096 + 555, 2), (3 * 5000, 3)])
def test_single_collective_with_boundary_blocks_of_several_tiles(n, world):
    """Capacity of several tiles per boundary block: the prefix of the tiles before a block's own comes from
    k_propagate's sub-unit sums when the block is tile-aligned, else it is recomputed -- same bits either way."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(14, missing=0.1)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalComm(world))
    f.MIN_CAP = 3500                      # rounded up to 4096 = 4 tiles
    ll, ess = f.ll_filter(t, y, has)
    assert f.last_single and f.last_attempts == 1 and f.last_cap >= 3500
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("world", [3, 4, 8])
@pytest.mark.parametrize("name,n,T", [("c2_model", 20000, 30), ("c3_model", 9000, 16)])
def test_trimmed_exchange_reads_only_headers_of_non_adjacent_segments(world, name, n, T):
    """The library's all-to-all-v (single_collective = 2) moves whole segments between adjacent ranks only and 12 header
    words between every other pair.  LocalCommTrimmed moves exactly that and fills the rest of every receive segment
    with NaN: if k_offspring_expand_spec (or the next k_propagate) read anything else, the oracle's bits could not
    come out."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalCommTrimmed
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(T, missing=0.1)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    f = ShardedFilter(shards, LocalCommTrimmed(world))
    ll, ess = f.ll_filter(t, y, has)
    assert f.last_single and f.last_attempts == 1
    oll, oess, opart = _oracle_run(model, n, t, y, has)
    assert (ll, ess) == (oll, oess[-1])
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.parametrize("peer", [False, True])
@pytest.mark.parametrize("world,name,n", [(2, "c2_model", 5000), (4, "c1_model", 6000), (8, "c2_model", 16000), (3, "c3_model", 4100)])
def test_local_shards_stratified_resampling(world, name, n, peer):
    """Resampling.stratifiedResampling (model/Resampling.scala:78-86) over shards: one uniform per slot from the Philox stream of the
    GLOBAL slot, so the ancestors of a rank's slots are again its own particles plus its neighbours' boundary blocks -- the same
    exchange, the stratified slot counts.  Bit-identical to the single-rank oracle with the same resampler: collective and peer-written
    exchange, the exact exchange, a continued series."""
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm, LocalCommPeer
    model = getattr(cases, name)()
    t, y, has = cases.poisson_counts(9, missing=0.15)
    shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
    for s in shards:
        s.set_option(2, 1)                      # CSSM_OPT_RESAMPLER = CSSM_RESAMPLE_STRATIFIED
    f = ShardedFilter(shards, (LocalCommPeer if peer else LocalComm)(world))
    o = oracle.OraclePf(model.descriptor(), n, cases.SEED, oracle.RESAMPLE_STRATIFIED)
    oll, _, oess, _ = o.filter(t, y, has)
    for exact in ((False,) if peer else (False, True)):
        ll, ess = f.ll_filter(t, y, has, exact=exact)
        assert (ll, ess) == (oll, oess[-1]), (exact, ll, oll)
        np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), o.particles())
    f.ll_filter(t[:5], y[:5], has[:5])
    assert f.ll_filter_more(t[5:], y[5:], has[5:]) == (oll, oess[-1])
    with pytest.raises(Exception):
        shards[0].set_option(2, 2)              # multinomial: another exchange altogether, refused on a shard
    for s in shards:
        s.close()


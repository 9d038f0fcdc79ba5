"""The sharded exchange on the GROUP sums (Scalars::grp / grp2, round 5): a shard's propagate adds every unit's sum (and sum of squares) to
its group of 32 units; the header blocks of the exchange total 2 x 32 group sums instead of every unit sum, and the offspring blocks take
their prefix inside the rank from the group sums + their own group's units -- everything local is done before a block waits for the peers'
headers (k_exchange_offspring<..., GRP>, k_offspring_expand_spec<..., GRP>; ParticleFilter.scala:124-130, Resampling.scala:52-72).

Two kinds of cases: (1) shards of at least 64 units -- the size from which the library turns the group sums on -- with whole and ragged
boundary blocks, missing and outlying observations, capacity misses, continued series, both resamplers, the LGCP; (2) the existing
sharded and peer tests run AGAIN with CSSM_GRP_MIN_UNITS=1, which sends their small shards (a few units, partial groups, every world size)
through the same kernels.  Everything against the single-rank oracle bit for bit."""
import numpy as np
import pytest

import cases
from oracle import oracle

# (2) re-collected under this module's fixture; their parametrisations travel with the functions
from test_gpu_peer import (test_an_outlying_observation_is_redone_in_place_also_in_a_continued_series,   # noqa: F401
                           test_peer_exchange_lgcp, test_peer_exchange_local_shards_match_single_rank_oracle,
                           test_peer_exchange_resumes_capacity_misses_and_redoes_an_outlier_in_place,
                           test_peer_exchange_world1_under_torch_distributed)
from test_gpu_sharded import (test_continued_sharded_series_equals_the_whole_series,   # noqa: F401
                              test_lgcp_series_resumes_a_capacity_miss, test_local_shards_lgcp, test_local_shards_stratified_resampling,
                              test_series_resumes_a_capacity_miss_and_repeats_after_an_outlying_observation,
                              test_sharded_filter_returns_the_sampled_path, test_single_collective_series_matches_oracle,
                              test_single_collective_with_boundary_blocks_of_several_tiles)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _group_sums_from_one_unit(monkeypatch, request):
    if "native_threshold" not in request.keywords:
        monkeypatch.setenv("CSSM_GRP_MIN_UNITS", "1")


def _filters(model, n, world, peer, prec=0):
    from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
    from local_comm import LocalComm, LocalCommPeer
    shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=prec) for r in range(world)]
    return shards, ShardedFilter(shards, (LocalCommPeer if peer else LocalComm)(world))


def _oracle(model, n, t, y, has, prec=0, flags=0):
    o = oracle.OraclePf(model.descriptor(prec), n, cases.SEED, flags)
    ll, _, ess_t, _ = o.filter(t, y, has)
    return ll, int(ess_t[-1]), o.particles()


@pytest.mark.native_threshold
@pytest.mark.parametrize("peer", [True, False])
@pytest.mark.parametrize("world,n", [(2, 2 * 65536), (2, 140001), (3, 3 * 70000 + 17), (4, 4 * 66560)])
def test_shards_of_64_units_and_more_on_group_sums(world, n, peer):
    """The library's own threshold (no environment override): >= 64 units per rank.  140001 / 3 x 70000 + 17: the last rank is smaller and
    its last boundary block ragged (the header block's fallback path beside group-sum offspring blocks)."""
    model = cases.c2_model()
    t, y, has = cases.poisson_counts(9, missing=0.2)
    y = y.copy(); y[5] = 60.0                               # an outlying observation: redone in place (peer) / the series repeated (collective)
    shards, f = _filters(model, n, world, peer)
    ll, ess = f.ll_filter(t[:6], y[:6], has[:6])
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle(model, n, t, y, has)
    assert (ll, ess) == (oll, oess)
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.native_threshold
@pytest.mark.parametrize("peer", [True, False])
def test_group_sum_shards_resume_capacity_misses_and_continue(peer):
    """A capacity of one row per pair misses at 2 x 80 units; the misses are resumed in place (the redone exchange reads the same group
    sums: nobody cleared them), the series is continued, and the sets of group sums keep rotating in step with the exchanges."""
    from composablestatespacemodels_amd.sharded import ShardedFilter
    model, n, world = cases.c2_model(), 2 * 81920, 2
    t, y, has = cases.poisson_counts(14, missing=0.15)
    shards, f = _filters(model, n, world, peer)
    f.MIN_CAP, f.CAP_SQRT = 1, 0.0                         # one row per pair
    ll, ess = f.ll_filter(t[:8], y[:8], has[:8])
    assert f.last_resumes >= 1
    ll, ess = f.ll_filter_more(t[8:], y[8:], has[8:])
    oll, oess, opart = _oracle(model, n, t, y, has)
    assert (ll, ess) == (oll, oess)
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()
    assert ShardedFilter.MIN_CAP == 1024


@pytest.mark.native_threshold
@pytest.mark.parametrize("peer", [True, False])
def test_group_sum_shards_lgcp_and_stratified(peer):
    """The LGCP's generic propagate kernel (k_propagate<D, true, ..., SUMS = 2>) adds to the group sums as well; stratified resampling
    runs the same exchange kernels in their RS = 1 instantiation."""
    model, n, world = cases.c4_model(), 2 * 69632, 2
    t, y, has = cases.event_times(6, horizon=3.0)
    shards, f = _filters(model, n, world, peer, prec=2)
    ll, ess = f.ll_filter(t, y, has, lgcp=True)
    oll, oess, opart = _oracle(model, n, t, y, has, prec=2)
    assert (ll, ess) == (oll, oess)
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()
    model = cases.c1_model()
    t, y, has = cases.poisson_counts(7, missing=0.2)
    shards, f = _filters(model, 3 * 66000, 3, peer)
    for s in shards:
        s.set_option(2, 1)                                  # CSSM_OPT_RESAMPLER = stratified
    ll, ess = f.ll_filter(t, y, has)
    oll, oess, opart = _oracle(model, 3 * 66000, t, y, has, flags=oracle.RESAMPLE_STRATIFIED)
    assert (ll, ess) == (oll, oess)
    np.testing.assert_array_equal(np.concatenate([s.particles() for s in shards], axis=1), opart)
    for s in shards:
        s.close()


@pytest.mark.native_threshold
@pytest.mark.parametrize("split,peer", [("1", True), ("2", True), ("4", False), ("2", False)])
def test_lgcp_shards_with_several_propagate_blocks_per_unit(split, peer, monkeypatch):
    """An LGCP shard of 2^21 particles has units of two tiles; its propagate runs blocks of HALF a unit (CSSM_SHARD_LGCP_SPLIT = 2, the
    default; 4: a quarter), every block adding its sums to its unit's group (Scalars::grp / grp2) and the exchange's offspring blocks
    adding up their group's block sums: the same bits as whole-unit blocks and as the single-GPU handle (2 shards of 2^21 here)."""
    from composablestatespacemodels_amd.filter import NativePf
    monkeypatch.setenv("CSSM_SHARD_LGCP_SPLIT", split)
    model, n, world = cases.c4_model(), 1 << 22, 2
    t, y, has = cases.event_times(5, horizon=2.0)
    g = NativePf(model, n, cases.SEED, lgcp_precision=2)
    ll1, _, ess1, _ = g.run(t, y, has)
    ref = g.particles()
    g.close()
    shards, f = _filters(model, n, world, peer, prec=2)
    ll, ess = f.ll_filter(t[:3], y[:3], has[:3], lgcp=True)
    ll, ess = f.ll_filter_more(t[3:], y[3:], has[3:], lgcp=True)
    assert (ll, ess) == (ll1, int(ess1[-1]))
    for s in shards:
        np.testing.assert_array_equal(s.particles(), ref[:, s.first:s.first + s.n])
        s.close()

"""Rows the pack of the peer-written exchange sends per neighbour segment and observation (the eager rows, or the needed ones where they
are more), against the capacity of a boundary block (what CSSM_PEER_ALL_ROWS=1 sends), and how often rows beyond the eager ones were
needed: `world` shards of ONE process on this GPU (tests/local_comm.py), the bench workload (C2).  CSSM_PEER_EAGER_ROWS=1 shows the bare need.
usage: rows_sent.py [world=8] [particles per shard=1048576] [observations=100]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
from local_comm import LocalCommPeer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
n = world * per
model = cases.c2_model()
t, y, has = cases.poisson_counts(T, missing=0.05)
shards = [GpuShard(model, n, r, world, cases.SEED, 0) for r in range(world)]
f = ShardedFilter(shards, LocalCommPeer(world))
ll, ess = f.ll_filter(t, y, has)
assert f.last_peer, "the series did not run on the peer-written exchange"
rows = [s.peer_rows() for s in shards]
tot_rows, tot_seg, beyond = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows)
d = shards[0].d
row_bytes = (d + 2) * 8
cap = f.last_cap
print(f"world {world} x {per} particles, {int(np.count_nonzero(has))} weighted observations, ll {ll!r} ess {ess}, resumes {f.last_resumes}")
print(f"capacity {cap} rows per boundary block = {cap * row_bytes / 1024:.0f} KiB per neighbour and observation if every row travels")
print(f"rows written (eager = {os.environ.get('CSSM_PEER_EAGER_ROWS', '4096')}): {tot_rows} over {tot_seg} neighbour segments = {tot_rows / max(tot_seg, 1):.0f} rows = "
      f"{tot_rows / max(tot_seg, 1) * row_bytes / 1024:.1f} KiB per neighbour and observation ({100.0 * tot_rows / max(tot_seg * cap, 1):.1f} % of the capacity); "
      f"{beyond} segments ({100.0 * beyond / max(tot_seg, 1):.1f} %) needed rows beyond the eager ones")
for s in shards:
    s.close()

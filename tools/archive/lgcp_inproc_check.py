import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, cases
from oracle import oracle
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter
from local_comm import LocalCommPeer
world, n, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model = cases.c4_model(); t, y, has = cases.event_times(T, horizon=0.1 * T)
o = oracle.OraclePf(model.descriptor(2), n, cases.SEED); oll, _, oess, _ = o.filter(t, y, has)
shards = [GpuShard(model, n, r, world, cases.SEED, 0, lgcp_precision=2) for r in range(world)]
f = ShardedFilter(shards, LocalCommPeer(world))
cut = T // 3
f.ll_filter(t[:cut], y[:cut], has[:cut], lgcp=True)
ll, ess = f.ll_filter_more(t[cut:], y[cut:], has[cut:], lgcp=True)
print("in-process world", world, "N", n, "T", T, "eager", os.environ.get("CSSM_PEER_EAGER_ROWS"), ":", (ll, ess) == (oll, int(oess[-1])), ll, oll, ess, int(oess[-1]), "peer", f.last_peer, "resumes", f.last_resumes, f.last_level_redos)

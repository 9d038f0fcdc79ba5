#!/usr/bin/env python3
"""Host-side anatomy of a continued K-observation leg of the sharded filter at world = 1 (bench.py --sharded --steps K): wall time of the
leg against K x the steady-state step, and the host time of the three library calls a leg is made of.  usage: shard_leg_probe.py [K] [N]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch, torch.distributed as dist, cases
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29549"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = cases.c2_model(); t, y, has = cases.poisson_counts(8 + 40 * K + 600)
shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
f.ll_filter(t[:8], y[:8], has[:8])
lo = 8
f.ll_filter_more(t[lo:lo + 500], y[lo:lo + 500], has[lo:lo + 500]); lo += 500
torch.cuda.synchronize(); t0 = time.perf_counter()
f.ll_filter_more(t[lo:lo + 500 if False else lo + 100], y[lo:lo + 100], has[lo:lo + 100]); torch.cuda.synchronize()
step_us = (time.perf_counter() - t0) / 100 * 1e6; lo += 100
acc = {}
def wrap(name):
    fn = getattr(GpuShard, name)
    def w(self, *a, **k):
        t0 = time.perf_counter(); r = fn(self, *a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6); return r
    setattr(GpuShard, name, w)
for nm in ("begin_more", "series_peer", "status"):
    wrap(nm)
walls = []
for r in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.ll_filter_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K]); torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e6); lo += K
print(f"N={n} K={K}: leg wall median {np.median(walls):.1f} us = {np.median(walls)/K:.2f} us/step; 100-step leg {step_us:.2f} us/step; "
      f"host us per call (median): " + ", ".join(f"{k} {np.median(v):.1f}" for k, v in acc.items()) +
      f"; sum {sum(np.median(v) for v in acc.values()):.1f}; python around them {np.median(walls) - sum(np.median(v) for v in acc.values()):.1f}")
shard.close(); dist.destroy_process_group()

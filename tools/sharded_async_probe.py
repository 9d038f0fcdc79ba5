"""What would the sharded path cost at world = 1 without the per-step host read?  (experiment; run on the GPU box)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.distributed as dist, cases
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29545"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = cases.c2_model(); t, y, has = cases.poisson_counts(220)
for n in (1 << 20, 1 << 24):
    for mode in ("sync", "nosync", "nosync_nocoll", "kernels_only"):
        shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
        f = ShardedFilter([shard], DistComm())
        f.init(0.0)
        dummy = shard.buffer("recv", 8)
        def step(s):
            if mode == "sync":
                f.step(float(t[s]), float(y[s]), True); return
            shard.propagate(float(t[s]), float(y[s]), 1)
            if mode in ("nosync",):
                f.comm.all_gather([shard.all_sums], [shard.sums5])
            else:
                shard.all_sums.copy_(shard.sums5)
            shard.offspring()
            if mode == "nosync":
                f.comm.all_to_all_counts([shard.recv_count], [shard.send_count])
            if mode != "kernels_only":
                shard.adopt(dummy, 0, 0, 0, n)
            else:
                shard.adopt(dummy, 0, 0, 0, n)
        for s in range(20): step(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(20, 220): step(s)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"sharded world=1 N={n} mode={mode}: {w/200*1e6:.1f} us/step  ll={shard.result()[0]:.6f}", flush=True)
        shard.close()
dist.destroy_process_group()

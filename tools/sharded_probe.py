"""Time the sharded (RCCL) code path at world = 1 against the single-GPU path (run on the GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.distributed as dist, cases
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29544"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = cases.c2_model(); t, y, has = cases.poisson_counts(220)
for n in (1 << 20, 1 << 24):
    shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
    f = ShardedFilter([shard], DistComm())
    f.init(0.0)
    for s in range(20): f.step(float(t[s]), float(y[s]), True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(20, 220): f.step(float(t[s]), float(y[s]), True)
    torch.cuda.synchronize(); w = time.perf_counter() - t0
    print(f"sharded RCCL path, world=1, N={n}: {w/200*1e6:.1f} us/step, {n*200/w/1e9:.2f} G particle-steps/s", flush=True)
    shard.close()
dist.destroy_process_group()

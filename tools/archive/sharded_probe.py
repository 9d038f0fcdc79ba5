"""Time the sharded (RCCL) code path at world = 1 against the single-GPU path (run on the GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, torch.distributed as dist, cases
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29544"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = cases.c2_model(); t, y, has = cases.poisson_counts(220)
for n in ([int(a) for a in sys.argv[1:]] or [1 << 20, 1 << 24]):
    shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
    f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
    for exact in ((False,) if os.environ.get("PROBE_FIXED_ONLY") else (True, False)):
        f.ll_filter(t[:20], y[:20], has[:20], exact=exact)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ll, ess = f.ll_filter(t[:200], y[:200], has[:200], exact=exact)
        torch.cuda.synchronize(); w = time.perf_counter() - t0
        print(f"sharded RCCL path, world=1, N={n}, {'exact (host-read)' if exact else 'single-collective'} exchange: {w/200*1e6:.1f} us/step, "
              f"{n*200/w/1e9:.2f} G particle-steps/s  cap={f.last_cap} attempts={f.last_attempts} ll={ll:.6f}", flush=True)
    shard.close()
dist.destroy_process_group()

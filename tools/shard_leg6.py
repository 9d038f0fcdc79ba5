#!/usr/bin/env python3
"""(round 6) A continued K-observation leg of the sharded filter at world 1 (bench.py --sharded --steps K): wall, the GPU's own clock from the
leg's first kernel to k_finish (cssm_pf_last_device_us), host time of the three library calls, Python around them.
usage: shard_leg6.py [K] [N] [model c2|c4]"""
import ctypes as C, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch, torch.distributed as dist, cases
from composablestatespacemodels_amd import _abi
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29549"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = cases.c2_model(); t, y, has = cases.poisson_counts(8 + 60 * K + 700)
shard = GpuShard(model, n, 0, 1, cases.SEED, 0)
f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
f.ll_filter(t[:8], y[:8], has[:8])
lo = 8
def dev_us():
    us = C.c_double(); _abi.check(shard.lib.cssm_pf_last_device_us(shard._h, C.byref(us))); return us.value
f.ll_filter_more(t[lo:lo + 500], y[lo:lo + 500], has[lo:lo + 500]); lo += 500
torch.cuda.synchronize(); t0 = time.perf_counter()
f.ll_filter_more(t[lo:lo + 100], y[lo:lo + 100], has[lo:lo + 100]); torch.cuda.synchronize()
step_us = (time.perf_counter() - t0) / 100 * 1e6; lo += 100
print(f"100-step leg: wall {step_us:.2f} us/step, device {dev_us() / 100:.2f} us/step")
acc = {}
def wrap(name):
    fn = getattr(GpuShard, name)
    def w(self, *a, **k):
        t0 = time.perf_counter(); r = fn(self, *a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6); return r
    setattr(GpuShard, name, w)
for nm in ("begin_more", "series_peer", "status"):
    wrap(nm)
walls, devs, calls = [], [], []
MODE = os.environ.get("LEG_BRACKET", "sync")     # sync | barrier (dist.barrier() ahead of the synchronise, as bench.py brackets a leg) | barrier+touch
touch = torch.zeros(1, device="cuda")
for r in range(40):
    if MODE != "sync":
        dist.barrier()
    if MODE == "barrier+touch":
        torch.cuda.synchronize(); touch.add_(1.0)      # (one launch on the filter's stream behind the collective's)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f.ll_filter_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K]); t1 = time.perf_counter(); torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e6); calls.append((t1 - t0) * 1e6); devs.append(dev_us()); lo += K
m = np.median
print(f"[{MODE}] N={n} K={K}: leg wall {m(walls):.1f} us = {m(walls)/K:.2f} us/step; call {m(calls):.1f}; device {m(devs):.1f} = {m(devs)/K:.2f} us/step; "
      f"host us per call: " + ", ".join(f"{k} {m(v):.1f}" for k, v in acc.items()) +
      f"; sum {sum(m(v) for v in acc.values()):.1f}; python around them {m(calls) - sum(m(v) for v in acc.values()):.1f}")
print("device legs:", " ".join(f"{v:.0f}" for v in devs))
shard.close(); dist.destroy_process_group()

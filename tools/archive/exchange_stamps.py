#!/usr/bin/env python3
"""Where k_exchange_offspring's time goes (the merged pack + offspring launch of the peer-written exchange, world = 1 over RCCL):
clock stamps of a diagnostic build.

Build (in composablestatespacemodels_amd/csrc, after `make`):
    mkdir -p build_stamps
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mfma --offload-arch=gfx950 -Wno-unused-function -DCSSM_OFF_STAMPS -c -o build_stamps/shard.o cssm_shard.hip
    hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_stamps/libcssm_pf_stamps.so build/pf.o build_stamps/shard.o build/batch.o build/model.o build/rtc.o build/prop_d*.o -ldl
Run:   CSSM_PF_LIB=.../build_stamps/libcssm_pf_stamps.so python tools/archive/exchange_stamps.py [N] [lgcp]
Stamps (100 MHz): pack blocks -- 4 header block entered, 5 header written + flag released, 6 unit-sum prefixes written + announced;
offspring blocks -- 0 entry, 7 all flags seen, 1 headers in LDS + level checked, 2 own ancestors written, 3 received rows expanded."""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import cases  # noqa: E402
from composablestatespacemodels_amd import _abi  # noqa: E402
from composablestatespacemodels_amd.sharded import DistComm, GpuShard, ShardedFilter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
lgcp = len(sys.argv) > 2 and sys.argv[2] == "lgcp"
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29547"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lib = _abi.load_library()
lib.cssm_pf_debug_spec_stamps.restype = C.c_int
lib.cssm_pf_debug_spec_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
if lgcp:
    model = cases.c4_model(); t, y, has = cases.event_times(40, horizon=4.0)
else:
    model = cases.c2_model(); t, y, has = cases.poisson_counts(40)
shard = GpuShard(model, n, 0, 1, cases.SEED, 0, lgcp_precision=2 if lgcp else 0)
f = ShardedFilter([shard], DistComm(device=torch.device("cuda", 0)))
f.ll_filter(t, y, has, lgcp=lgcp)
assert f.last_peer, "the peer-written exchange did not run"
out = np.zeros(2048 * 8, dtype=np.uint64)
assert lib.cssm_pf_debug_spec_stamps(shard._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size) == 0
s = out.reshape(2048, 8).astype(np.int64)
off = s[s[:, 0] > 0]
t0 = min(off[:, 0].min(), s[s[:, 4] > 0][:, 4].min() if (s[:, 4] > 0).any() else 1 << 62)
print("N = %d, %s, %d offspring blocks, cap %d" % (n, "lgcp" if lgcp else "c2", len(off), f.last_cap))
for k, name in ((4, "pack: header block entered"), (5, "pack: header + flag out"), (6, "pack: unit prefixes announced")):
    v = s[s[:, k] > 0][:, k]
    if len(v):
        print("  %-34s %6.2f us" % (name, (v.min() - t0) * 0.01))
for k, name in ((0, "entry"), (7, "all flags seen"), (1, "headers in LDS, level ok"), (2, "own ancestors written"), (3, "received rows expanded")):
    v = (off[:, k] - t0) * 0.01
    print("  %-34s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
shard.close()
dist.destroy_process_group()

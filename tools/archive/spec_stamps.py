#!/usr/bin/env python3
"""Where k_offspring_expand_spec's time goes (sharded single-collective step, one rank): clock stamps of a diagnostic build.

Build: as tools/archive/offspring_stamps.py says, with cssm_shard.hip compiled -DCSSM_OFF_STAMPS into build_stamps/shard.o.
Run:   CSSM_PF_LIB=.../build_stamps/libcssm_pf_stamps.so python tools/spec_stamps.py
Stamps (100 MHz): 0 entry, 1 level + headers' verdict known, 2 own particles' ancestors written, 3 received rows expanded."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cases  # noqa: E402
from composablestatespacemodels_amd import _abi  # noqa: E402
from composablestatespacemodels_amd.sharded import GpuShard, ShardedFilter  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from local_comm import LocalComm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
lib = _abi.load_library()
lib.cssm_pf_debug_spec_stamps.restype = C.c_int
lib.cssm_pf_debug_spec_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
t, y, has = cases.poisson_counts(40)
shard = GpuShard(cases.c2_model(), n, 0, 1, cases.SEED, 0)
f = ShardedFilter([shard], LocalComm(1))
f.ll_filter(t, y, has)
nb = min((n + 1023) // 1024, 2048)
out = np.zeros(nb * 8, dtype=np.uint64)
assert lib.cssm_pf_debug_spec_stamps(shard._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size) == 0
s = out.reshape(nb, 8).astype(np.int64)
t0 = s[:, 0].min()
for k, name in ((0, "entry"), (1, "verdict known"), (2, "own ancestors written"), (3, "received rows expanded")):
    v = (s[:, k] - t0) * 0.01
    print("  %-24s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
shard.close()

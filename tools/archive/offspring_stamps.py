#!/usr/bin/env python3
"""Where k_offspring_self's time goes at the bench size: clock stamps left by every block of a diagnostic build.

Build (in composablestatespacemodels_amd/csrc, after `make`):
    mkdir -p build_stamps
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mfma --offload-arch=gfx950 -DCSSM_OFF_STAMPS -c -o build_stamps/pf.o cssm_pf.hip
    hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_stamps/libcssm_pf_stamps.so build_stamps/pf.o build/shard.o build/model.o build/prop_d*.o
Run on the GPU box:
    CSSM_PF_LIB=composablestatespacemodels_amd/csrc/build_stamps/libcssm_pf_stamps.so python tools/archive/offspring_stamps.py [N]

Stamps (100 MHz constant clock, 10 ns): 0 kernel entry, 1 behind the unit-sum scan (first loads have landed, one barrier),
2 weights on the 2^-96 grid + wave scan done, 3 behind the tile's barrier, 4 end slots known, 5 ancestors assembled and
stores issued, 7 publisher done.  Printed relative to the earliest entry of the launch: quantiles over the unit blocks.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cases  # noqa: E402
from composablestatespacemodels_amd import _abi  # noqa: E402
from composablestatespacemodels_amd.filter import NativePf  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    lib = _abi.load_library()
    if not hasattr(lib, "cssm_pf_debug_stamps"):
        try:
            lib.cssm_pf_debug_stamps
        except AttributeError:
            raise SystemExit("this library was not built with -DCSSM_OFF_STAMPS (see the docstring)")
    lib.cssm_pf_debug_stamps.restype = C.c_int
    lib.cssm_pf_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
    T = 60
    t, y, has = cases.poisson_counts(T)
    g = NativePf(cases.c2_model(), n, cases.SEED)
    g.run(t, y, has)   # warm
    g.run(t, y, has)   # the stamps of the LAST weighted observation remain
    nblocks = (n + 1023) // 1024 + 1
    out = np.zeros(nblocks * 8, dtype=np.uint64)
    rc = lib.cssm_pf_debug_stamps(g._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size)
    assert rc == 0, rc
    s = out.reshape(nblocks, 8).astype(np.int64)
    t0 = s[:, 0].min()
    units, pub = s[:-1], s[-1]
    names = {0: "entry", 1: "behind unit-sum scan", 2: "grid weights + wave scan", 3: "behind tile barrier", 4: "end slots known",
             5: "ancestors stored"}
    print("N = %d, %d unit blocks + publisher; times in us after the launch's earliest block entry" % (n, nblocks - 1))
    for k, name in names.items():
        v = (units[:, k] - t0) * 0.01
        print("  %-26s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (name, v.min(), np.percentile(v, 10), np.median(v),
                                                                               np.percentile(v, 90), v.max()))
    d = (units[:, 5] - units[:, 0]) * 0.01
    print("  block lifetime (0 -> 5)    min %6.2f  median %6.2f  max %6.2f" % (d.min(), np.median(d), d.max()))
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5)):
        v = (units[:, b] - units[:, a]) * 0.01
        print("  phase %d -> %d               median %6.2f  p90 %6.2f" % (a, b, np.median(v), np.percentile(v, 90)))
    # who is late?  (blocks go round-robin over the 8 XCDs; a CU holds 4-5 blocks)
    late = (units[:, 1] - t0) * 0.01
    print("  behind-scan time by XCD (block %% 8): " + " ".join("%.2f" % np.median(late[x::8]) for x in range(8)))
    q = len(late) // 8
    print("  behind-scan time by block index octile: " + " ".join("%.2f" % np.median(late[i * q:(i + 1) * q]) for i in range(8)))
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "stamps.npy"), s)
    print("  publisher: entry %.2f, behind scan %.2f, done %.2f" % ((pub[0] - t0) * 0.01, (pub[1] - t0) * 0.01, (pub[7] - t0) * 0.01))
    g.close()


if __name__ == "__main__":
    main()

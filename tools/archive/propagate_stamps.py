#!/usr/bin/env python3
"""Where k_propagate_self's time goes at the bench size: clock stamps left by every block of a diagnostic build.

Build (in composablestatespacemodels_amd/csrc, after `make`):
    mkdir -p build_stamps
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mfma --offload-arch=gfx950 -DCSSM_PROP_STAMPS -DCSSM_PROP_D=3 -c -o build_stamps/prop_d3.o cssm_prop.hip
    hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_stamps/libcssm_pf_pstamps.so build/pf.o build/shard.o build/model.o \
        build_stamps/prop_d3.o $(ls build/prop_d*.o | grep -v 'prop_d3\\.o')
Run on the GPU box:
    CSSM_PF_LIB=composablestatespacemodels_amd/csrc/build_stamps/libcssm_pf_pstamps.so python tools/archive/propagate_stamps.py [N]

Stamps (100 MHz constant clock, 10 ns), wave 0 of every block: 0 kernel entry, 1 behind the table's barrier (first loads landed, the
first tile's Philox blocks drawn), 2 the first tile's normals drawn (its rows were requested before them), 3 its rows landed,
4 first tile computed and its stores issued, 5 second tile's rows landed (its normals drawn), 6 last tile computed, 7 block done
(reductions, sums stored, max published).  Printed relative to the earliest entry of the launch: quantiles over the blocks.
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
    try:
        fn = lib.cssm_prop_debug_stamps_d3
    except AttributeError:
        raise SystemExit("this library was not built with -DCSSM_PROP_STAMPS (see the docstring)")
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
    T = 60
    t, y, has = cases.poisson_counts(T)
    g = NativePf(cases.c2_model(), n, cases.SEED)
    g.run(t, y, has)   # warm
    g.run(t, y, has)   # the stamps of the LAST observation remain
    nblocks = min(8192, (n + 1023) // 1024 if n >= (1 << 20) else (n + 511) // 512)
    out = np.zeros(nblocks * 8, dtype=np.uint64)
    rc = fn(out.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.size)
    assert rc == 0, rc
    s = out.reshape(nblocks, 8).astype(np.int64)
    t0 = s[:, 0].min()
    names = {0: "entry", 1: "behind the table's barrier", 2: "first tile's normals drawn", 3: "its rows landed", 4: "first tile computed",
             5: "second tile's rows landed", 6: "last tile computed", 7: "block done"}
    print("N = %d, %d blocks; times in us after the launch's earliest block entry" % (n, nblocks))
    for k, name in names.items():
        v = (s[:, k] - t0) * 0.01
        print("  %-28s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (name, v.min(), np.percentile(v, 10), np.median(v),
                                                                                 np.percentile(v, 90), v.max()))
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)):
        v = (s[:, b] - s[:, a]) * 0.01
        print("  phase %d -> %d                 median %6.2f  p10 %6.2f  p90 %6.2f" % (a, b, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
    d = (s[:, 7] - s[:, 0]) * 0.01
    print("  block lifetime (0 -> 7)      min %6.2f  median %6.2f  max %6.2f" % (d.min(), np.median(d), d.max()))
    g.close()


if __name__ == "__main__":
    main()

"""Whole-step device time (separate and fused sums) of alternative builds, same box, interleaved repetitions."""
import glob, os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(R, "composablestatespacemodels_amd", "csrc", "libcssm_pf*.so")))
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import composablestatespacemodels_amd._abi as abi
abi.LIB_PATH = %r
import cases
from composablestatespacemodels_amd.filter import NativePf
m = cases.c2_model(); t, y, has = cases.poisson_counts(300)
out = []
for n in (1 << 20, 1 << 24):
    for fused in (0, 1):
        pf = NativePf(m, n, 1); pf.set_option(3, fused); T = 300 if n < (1 << 22) else 60
        pf.run(t[:20], y[:20], has[:20])
        best = 1e9
        for _ in range(4):
            ll = pf.run(t[:T], y[:T], has[:T])[0]; best = min(best, pf.last_loop_ms() / T * 1e3)
        out.append("N=2^%%d %%s %%.1f (ll %%r)" %% (n.bit_length() - 1, "fused" if fused else "separate", best, ll)); pf.close()
print(os.path.basename(%r), " | ".join(out), flush=True)
'''
for rep in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", code % (R, R, lib, lib)])

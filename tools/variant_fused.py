"""Time the fused-sums path (what sharded handles run) of alternative builds: k_propagate<..., SUMS=true> + k_offspring."""
import glob, os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(R, "composablestatespacemodels_amd", "csrc", "libcssm_pf*.so")))
code = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import composablestatespacemodels_amd._abi as abi
abi.LIB_PATH = %r
import cases
from composablestatespacemodels_amd.filter import NativePf
for n in (1 << 20, 1 << 24):
    m = cases.c2_model(); t, y, has = cases.poisson_counts(220)
    pf = NativePf(m, n, 1); pf.set_option(3, 1); pf.run(t[:20], y[:20], has[:20])
    pf.run(t[:200], y[:200], has[:200]); loop = pf.last_loop_ms() / 200 * 1e3
    pf.profile(True); pf.run(t[:60], y[:60], has[:60]); p = pf.profile_read(); pf.close()
    print(os.path.basename(%r), "fused N=%%d: %%.1f us/step" %% (n, loop), {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in p.items() if v[1]}, flush=True)
'''
for lib in libs:
    subprocess.run([sys.executable, "-c", code % (R, R, lib, lib)])

"""Time k_propagate of alternative builds of the library (libcssm_pf_<tag>.so next to the default one)."""
import glob, os, sys, json, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(R, "composablestatespacemodels_amd", "csrc", "libcssm_pf*.so")))
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import composablestatespacemodels_amd._abi as abi
abi.LIB_PATH = %r
import cases
from composablestatespacemodels_amd.filter import NativePf
for n, d_model in ((1 << 24, cases.c2_model), (1 << 20, cases.c2_model), (1 << 22, cases.c3_model), (1 << 24, cases.c1_model)):
    m = d_model(); t, y, has = cases.poisson_counts(60)
    pf = NativePf(m, n, 1); pf.run(t[:10], y[:10], has[:10])
    pf.profile(True); pf.run(t, y, has); p = pf.profile_read(); pf.close()
    print(os.path.basename(%r), "N=%%d d=%%d" %% (n, pf.d), {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in p.items() if v[1]}, flush=True)
'''
for lib in libs:
    subprocess.run([sys.executable, "-c", code % (R, R, lib, lib)])

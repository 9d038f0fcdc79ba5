"""Run one batch series per child process (so that a GPU exception in one does not end the others) and report."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, faulthandler; faulthandler.enable()
sys.path.insert(0, '@ROOT@'); sys.path.insert(0, '@ROOT@/tests')
import numpy as np, cases
from composablestatespacemodels_amd.filter import NativePf
name, n, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
model = cases.dim_model(int(name[1:])) if name.startswith("d") else getattr(cases, name)()
t, y, has = cases.poisson_counts(T, missing=0.15)
g = NativePf(model, n, cases.SEED)
print("created", flush=True)
g.set_option(4, 1)
a = g.run(t[:1], y[:1], has[:1]); aa = g.ancestors().astype(np.int64)
print("series T=1 ll", a[0], "anc min/max", aa.min(), aa.max(), "bad count", int(np.sum(aa >= n)), flush=True)
bad = np.nonzero(aa >= n)[0]
print("bad slots", bad[:12], "values", [hex(int(v)) for v in aa[bad[:12]]], flush=True)
g2 = NativePf(model, n, cases.SEED); g2.set_option(4, 0)
b = g2.run(t[:1], y[:1], has[:1]); ab = g2.ancestors().astype(np.int64)
print("plain  T=1 ll", b[0], "anc equal", np.array_equal(aa, ab), "n diff", int(np.sum(aa != ab)), "first diffs", np.nonzero(aa != ab)[0][:12], flush=True)
'''.replace('@ROOT@', ROOT)
for spec in sys.argv[1:]:
    name, n, T = spec.split(":")
    r = subprocess.run([sys.executable, "-c", CHILD, name, n, T], capture_output=True, text=True, timeout=120)
    print(spec, "rc", r.returncode, "|", r.stdout.strip().replace("\n", " ; "), "|", r.stderr.strip()[-int(os.environ.get("PROBE_TAIL", "300")):].replace("\n", " ; "))

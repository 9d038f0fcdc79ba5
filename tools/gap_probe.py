"""A plain batch run for rocprofv3 --kernel-trace: usage gap_probe.py N one_launch [T=300] [model=c2]."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import cases
from composablestatespacemodels_amd.filter import NativePf
n = int(sys.argv[1]); one = int(sys.argv[2]); T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
model = cases.c1_model() if (len(sys.argv) > 4 and sys.argv[4] == "c1") else cases.c2_model()
t, y, has = cases.poisson_counts(T)
g = NativePf(model, n, cases.SEED)
g.set_option(5, one)
for _ in range(3):
    g.run(t, y, has)
print(n, one, g.last_loop_ms() * 1e3 / T, "us per observation")
g.close()

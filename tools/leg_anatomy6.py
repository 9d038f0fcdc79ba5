"""(round 6) Anatomy of a continued 20-observation leg: the library's own phase clock (CSSM_CALL_TIMING: records built / upload enqueued /
all launches enqueued / results back, microseconds after the call's entry), the leg's DEVICE time from the GPU's clock stamps
(cssm_pf_last_device_us) and the host's wall around the call and around the closing synchronise.  usage (GPU box): python tools/leg_anatomy6.py [K]"""
import os, re, sys, time, tempfile
os.environ["CSSM_CALL_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, cases, torch
from composablestatespacemodels_amd.filter import NativePf
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t, y, has = cases.poisson_counts(60 * K + 10)
g = NativePf(cases.c2_model(), 1 << 20, cases.SEED)
g.run(t[:5], y[:5], has[:5])
cap = tempfile.TemporaryFile(mode="w+")
saved = os.dup(2); os.dup2(cap.fileno(), 2)
call, wall, dev = [], [], []
for r in range(50):
    lo = 5 + r * K
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_more(t[lo:lo + K], y[lo:lo + K], has[lo:lo + K])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    call.append((t1 - t0) * 1e6); wall.append((t2 - t0) * 1e6); dev.append(g.last_device_us())
os.dup2(saved, 2); cap.seek(0)
ph = [list(map(float, re.findall(r"records built ([\d.]+) us, upload enqueued ([\d.]+), \d+ steps enqueued ([\d.]+), results back ([\d.]+)", ln)[0])) for ln in cap if "cssm call" in ln]
ph = np.array(ph[-40:]); m = np.median
print(f"K={K}, median of the last 40 legs (us): records built {m(ph[:,0]):.1f} | upload enqueued {m(ph[:,1]):.1f} | all launches enqueued {m(ph[:,2]):.1f} | results back (library) {m(ph[:,3]):.1f}")
print(f"   call (Python) {m(call[-40:]):.1f} | + closing synchronise {m(wall[-40:]):.1f} | device (first kernel -> k_finish) {m(dev[-40:]):.1f}")
print(f"   library results-back minus device = {m(ph[:,3]) - m(dev[-40:]):.1f} (entry -> first kernel's first instruction, + k_finish's tail -> the host sees the word); Python around the library {m(call[-40:]) - m(ph[:,3]):.1f}; synchronise {m(wall[-40:]) - m(call[-40:]):.1f}")
g.close()

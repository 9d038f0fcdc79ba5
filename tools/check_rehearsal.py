"""Check a `bench.py --gpus N --backend gloo-gpu` line (the N-rank path rehearsed with ranks sharing one GPU) against a single-GPU handle of the same
total cloud: ll and ESS after the last timed leg must be identical.  usage (GPU box): python tools/check_rehearsal.py <line.json>
(At 2^20 particles per rank four ranks' exchange kernels do not fit one GPU side by side: the peer protocol then times out in the pre-flight and
the walk moves on to the collectives -- the behaviour wanted of it; on a node every rank has a GPU of its own.)"""
import json, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import bench
from composablestatespacemodels_amd.filter import NativePf
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
K, W, Rr = j["steps"], max(j["warmup"], 8), j["repeats"]
n = j["config"]["particles_total"]
model, t, y, has = bench.build_workload(W + (Rr + 1) * K, "c2")
T = W + Rr * K
g = NativePf(model, n, 20260101)
ll, _, ess, _ = g.run(t[:T], y[:T], has[:T])
print("sharded ll", j["ll"], "ess", j["ess_last"], "| single-GPU handle of N =", n, "ll", ll, "ess", int(ess[-1]), "| identical:", j["ll"] == ll and j["ess_last"] == int(ess[-1]))
print("chosen", j["exchange"]["chosen"], [ (s["protocol"], s["ok"]) for s in j["exchange"]["preflight"]])

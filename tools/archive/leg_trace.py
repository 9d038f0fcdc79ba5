"""Per-position kernel times inside short continued legs, from a rocprofv3 --kernel-trace CSV of tools/archive/leg_overhead.py:
usage  leg_trace.py <kernel_trace.csv> [K]   -- prints, for step index 0 .. K-1 of a leg, the median duration of k_propagate and
k_offspring and the median gap before them (a leg = the kernels between two k_finish launches)."""
import csv, sys, statistics, re
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
legs, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "k_finish" in name:
        if cur:
            legs.append(cur)
        cur = []
    elif "k_propagate" in name or "k_offspring" in name:
        cur.append((("P" if "k_propagate" in name else "O"), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
legs = [l for l in legs if len(l) == 2 * K][3:]   # whole legs, the first few dropped
print(f"{len(legs)} legs of {K} steps")
for s in range(K):
    pd = [ (l[2*s][2] - l[2*s][1]) / 1e3 for l in legs]
    od = [ (l[2*s+1][2] - l[2*s+1][1]) / 1e3 for l in legs]
    g1 = [ (l[2*s+1][1] - l[2*s][2]) / 1e3 for l in legs]
    g0 = [ (l[2*s][1] - l[2*s-1][2]) / 1e3 for l in legs] if s else [0.0]
    print(f"step {s:2d}: k_propagate {statistics.median(pd):6.2f} us  k_offspring {statistics.median(od):6.2f} us  gap before P {statistics.median(g0):5.2f}  gap P->O {statistics.median(g1):5.2f}")
tot = [ (l[-1][2] - l[0][1]) / 1e3 for l in legs]
print(f"first kernel start -> last kernel end: median {statistics.median(tot):.1f} us = {statistics.median(tot)/K:.2f} us per step")

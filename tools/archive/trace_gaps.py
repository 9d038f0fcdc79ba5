"""Durations of and gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: usage trace_gaps.py <kernel_trace.csv>
prints, per kernel name: launches, median duration, median gap to the previous kernel's end (us)."""
import csv, sys, statistics, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = {}, {}
prev_end = None
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(name, []).append((e - s) / 1e3)
    if prev_end is not None:
        gap.setdefault(name, []).append((s - prev_end) / 1e3)
    prev_end = e
for k in dur:
    g = gap.get(k, [0.0])
    print(f"{k:62s} n={len(dur[k]):6d} dur med {statistics.median(dur[k]):7.2f} us  gap-before med {statistics.median(g):7.2f} us")

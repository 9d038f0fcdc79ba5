"""GPU timeline of short continued legs from a rocprofv3 run with --kernel-trace --memory-copy-trace:
   python tools/leg_timeline.py <dir>   -> per leg: copy start, copy end, first kernel start, last kernel end (us, relative)."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# a leg = events separated from the previous one by more than 15 us of idle time
legs = []; cur = []
for e in ev:
    if cur and e[0] - max(x[1] for x in cur) > 15000:
        legs.append(cur); cur = []
    cur.append(e)
if cur: legs.append(cur)
for leg in legs[-8:]:
    t0 = leg[0][0]
    print(f"leg of {len(leg)} events, {(max(x[1] for x in leg) - t0) / 1e3:.1f} us:")
    for s, e, n in leg[:4] + leg[-3:]:
        print(f"   {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {n}")

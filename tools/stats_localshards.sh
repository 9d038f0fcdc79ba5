#!/bin/bash
# rocprofv3 kernel statistics of 8 local shards of 2^20 particles (the kernels of an 8-rank run at its real capacity;
# the collectives are tensor copies here).  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_local8 -- python3 $R/tools/need_probe.py default > $R/gpurun_out/stats_local8.log 2>&1
cd $R
f=$(find gpurun_out/stats_local8 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['Percentage']) > 0.5:
        print(f"{row['Name'][:64]:64s} calls {row['Calls']:>6s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY

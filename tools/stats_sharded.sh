#!/bin/bash
# rocprofv3 kernel statistics of the sharded (RCCL, world = 1) path (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_sharded -- python3 $R/tools/sharded_probe.py > $R/gpurun_out/stats_sharded.log 2>&1
cd $R
grep sharded gpurun_out/stats_sharded.log
f=$(find gpurun_out/stats_sharded -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['Percentage']) > 0.3:
        print(f"{row['Name'][:70]:70s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us  total {float(row['TotalDurationNs'])/1e6:8.1f} ms")
PY

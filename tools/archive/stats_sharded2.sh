#!/bin/bash
# rocprofv3 kernel statistics of the sharded code path at world = 1 (run on the GPU box from the repo root).
TAG=${1:-s2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export PROBE_FIXED_ONLY=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_sh_${TAG} -- python3 $R/tools/archive/sharded_probe.py 1048576 > $R/gpurun_out/stats_sh_${TAG}.log 2>&1
cd $R
f=$(find gpurun_out/stats_sh_${TAG} -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['Percentage']) > 0.3:
        print(f"{row['Name'][:70]:70s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY

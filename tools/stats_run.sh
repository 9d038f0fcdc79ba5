#!/bin/bash
# rocprofv3 kernel-trace statistics of bench.py (run on the GPU box from the repo root): the bench configuration, the same
# model at N = 2^24 and Poisson-Brownian (d = 1) at N = 2^24.
# usage: tools/stats_run.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG}_1M -- python3 $R/bench.py --steps 500 --warmup 20 --no-cpu --no-16m --no-pmc > $R/gpurun_out/stats_${TAG}_1M.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG}_16M -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu --no-16m --no-pmc --particles 16777216 > $R/gpurun_out/stats_${TAG}_16M.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG}_16M_d1 -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu --no-16m --no-pmc --particles 16777216 --model c1 > $R/gpurun_out/stats_${TAG}_16M_d1.log 2>&1
cd $R
for s in 1M 16M 16M_d1; do
  f=$(find gpurun_out/stats_${TAG}_$s -name '*kernel_stats.csv' | head -1)
  echo "== $s: $f"; python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['Percentage']) > 0.5:
        print(f"{row['Name'][:60]:60s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us min {float(row['MinNs'])/1e3:8.1f}")
PY
  grep '^{' gpurun_out/stats_${TAG}_$s.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'k_propagate_us(events)', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done

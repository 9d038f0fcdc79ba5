#!/bin/bash
# rocprofv3 kernel-trace statistics of the sharded code path at world = 1 as bench.py --sharded runs it (peer-written exchange):
# configs[1] at 2^20 particles per rank and configs[3] (LGCP) at 2^21 per rank.  usage (GPU box, repo root): tools/stats_sharded_bench.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG}_sh_c2 -- python3 $R/bench.py --sharded --steps 500 --warmup 20 --no-cpu > $R/gpurun_out/stats_${TAG}_sh_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${TAG}_sh_c4 -- python3 $R/bench.py --sharded --model c4 --particles 2097152 --steps 200 --warmup 20 --no-cpu > $R/gpurun_out/stats_${TAG}_sh_c4.log 2>&1
cd $R
for s in sh_c2 sh_c4; do
  f=$(find gpurun_out/stats_${TAG}_$s -name '*kernel_stats.csv' | head -1)
  echo "== $s: $f"; python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['Percentage']) > 0.5:
        print(f"{row['Name'][:70]:70s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us min {float(row['MinNs'])/1e3:8.1f}")
PY
  grep '^{' gpurun_out/stats_${TAG}_$s.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('us_per_step', d['ms_per_step']*1e3, 'value', d['value'], d['exchange']['chosen'])"
done

#!/bin/bash
# A/B of bench.py switches on the GPU box: tools/ab_bench.sh <tag> "<args A>" "<args B>" ...
# Each variant runs at N = 2^20 (500 steps) and N = 2^24 (100 steps); one line per run in gpurun_out/ab_<tag>.log
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_${TAG}.log
: > $OUT
for v in "$@"; do
  for sz in "--steps 500 --warmup 20" "--steps 100 --warmup 5 --particles 16777216"; do
    echo "## $v $sz" >> $OUT
    timeout -k 10 300 python3 $R/bench.py --no-cpu $sz $v 2>/dev/null | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print(json.dumps({'value': d['value'], 'us_per_step': d['ms_per_step'] * 1e3, 'kernels_us': {k: round(v, 1) for k, v in d['kernels_us'].items() if v}, 'frac': round(d['roofline']['frac'], 3), 'll': d['ll']}))
except Exception as e:
    print('FAILED', e)
" >> $OUT
  done
done
cat $OUT

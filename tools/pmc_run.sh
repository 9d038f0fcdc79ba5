#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (run on the GPU box from the repo root).
# usage: tools/pmc_run.sh <tag> <particles> <steps> [extra bench.py flags, e.g. '--model c1']
set -u
TAG=${1:-r02}; N=${2:-16777216}; K=${3:-20}; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name} -- python3 $R/bench.py --steps $K --warmup 2 --no-cpu --no-16m --no-pmc --repeats 1 --particles $N $EXTRA > $R/gpurun_out/pmc_${TAG}_${name}.log 2>&1
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
cd $R
python3 tools/pmc_summary.py gpurun_out $TAG

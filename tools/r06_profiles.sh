#!/bin/bash
# Round 6's measurement session in one call (GPU box, repo root): bench lines, rocprofv3 kernel statistics, PMC passes, the sharded path at
# world 1, all five BASELINE configurations.  Outputs under gpurun_out/; copy_r06.sh (below) files the summaries into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
echo "== bench lines"
python3 bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; echo "default rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line_steps20.json 2> gpurun_out/r06_bench_line_steps20.err; echo "steps20 rc=$?"
python3 bench.py --model c4 --steps 100 --warmup 10 --no-cpu > gpurun_out/r06_bench_line_c4_lgcp_single_gpu.json 2>/dev/null; echo "c4 rc=$?"
echo "== sharded, world 1"
python3 bench.py --sharded --steps 500 --warmup 20 > gpurun_out/r06_sharded_world1_c2.json 2>/dev/null; echo "sh c2 rc=$?"
python3 bench.py --sharded --steps 20 --warmup 5 > gpurun_out/r06_sharded_world1_c2_steps20.json 2>/dev/null; echo "sh c2 k20 rc=$?"
python3 bench.py --sharded --model c4 --particles 2097152 --steps 200 --warmup 20 > gpurun_out/r06_sharded_world1_c4_lgcp.json 2>/dev/null; echo "sh c4 rc=$?"
echo "== kernel statistics"
bash tools/stats_run.sh r06 2>&1 | tail -14
bash tools/stats_sharded_bench.sh r06 2>&1 | tail -12
echo "== PMC"
bash tools/pmc_run.sh r06_1M 1048576 100 2>&1 | tail -4
bash tools/pmc_run.sh r06_16M 16777216 20 2>&1 | tail -4
bash tools/pmc_run.sh r06_16M_d1 16777216 20 '--model c1' 2>&1 | tail -4
echo "== configurations"
python3 tools/run_configs.py > gpurun_out/r06_configs.txt 2>&1; tail -8 gpurun_out/r06_configs.txt
echo "== done"

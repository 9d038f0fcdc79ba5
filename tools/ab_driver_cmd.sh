#!/bin/bash
# A/B of the runtime's completion-wait setting on the DRIVER'S EXACT bench command, interleaved on one box:
#   gpurun -- bash tools/ab_driver_cmd.sh [pairs]
# -> gpurun_out/ab_intr_<setting>_<i>.json (the bench lines), gpurun_out/ab_intr_summary.txt
P=${1:-3}
mkdir -p gpurun_out
for i in $(seq 1 $P); do
  for v in 0 1; do
    HSA_ENABLE_INTERRUPT=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/ab_intr_${v}_${i}.json 2> gpurun_out/ab_intr_${v}_${i}.err || exit 1
    echo "done $v $i"
  done
done
python3 - <<'PY' | tee gpurun_out/ab_intr_summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_intr_*_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    w = j["wall_ms_each"]
    print(f, "value %.3e" % j["value"], "us/step %.2f" % (j["ms_per_step"] * 1e3), "device_loop_ms %.4f" % j["device_loop_ms"],
          "legs", " ".join("%.3f" % x for x in w), "max/min %.3f" % (max(w) / min(w)), "dev_each", j.get("device_ms_each"))
PY

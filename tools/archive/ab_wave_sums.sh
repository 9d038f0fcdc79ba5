#!/bin/bash
# Same-box A/B of CSSM_WAVE_SUMS (k_offspring_self vs k_offspring_wave): bench.py's event-bracketed kernel times at 2^20 and 2^24
P=${1:-2}
mkdir -p gpurun_out
for i in $(seq 1 $P); do for v in 0 1; do
  CSSM_WAVE_SUMS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-generic > gpurun_out/abw_${v}_${i}.json 2> gpurun_out/abw_${v}_${i}.err || exit 1
done; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abw_*_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    r = j["roofline_16m"]
    print(f, "us/step %.2f" % (j["ms_per_step"] * 1e3), "loop %.2f" % (j["device_loop_ms"] * 1e3 / j["steps"]), "2^20:", {k: round(v, 2) for k, v in j["kernels_us"].items()},
          "2^24 d3:", {k: round(v, 1) for k, v in r["c2_d3"]["kernels_us"].items()}, "d1:", {k: round(v, 1) for k, v in r["c1_d1"]["kernels_us"].items()})
PY

#!/bin/bash
# Same-box A/B: the round-5 build (block-wide tiles, k_offspring_self: build_ab/libcssm_pf_nowr.so, built with EXTRA=-DCSSM_PROP_WR=0) against the
# current library with CSSM_WAVE_SUMS = 0 (same mapping through the new code), 1 (default: wave ranges beyond 2^20) and 2 (wave ranges everywhere)
mkdir -p gpurun_out
for i in 1 2 3; do
  CSSM_PF_LIB=$PWD/composablestatespacemodels_amd/csrc/build_ab/libcssm_pf_nowr.so python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-generic > gpurun_out/abx_r5build_$i.json 2>/dev/null || exit 1
  for v in 0 1 2; do CSSM_WAVE_SUMS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-generic > gpurun_out/abx_ws${v}_$i.json 2>/dev/null || exit 1; done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abx_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline_16m"]
    print(f, "us/step %.2f" % (j["ms_per_step"] * 1e3), "loop %.2f" % (j["device_loop_ms"] * 1e3 / j["steps"]), "2^20:", {k: round(v, 2) for k, v in j["kernels_us"].items()},
          "2^24 d3:", {k: round(v, 1) for k, v in r["c2_d3"]["kernels_us"].items()}, "step %.1f" % r["c2_d3"]["step_us"], "d1:", {k: round(v, 1) for k, v in r["c1_d1"]["kernels_us"].items()}, "step %.1f" % r["c1_d1"]["step_us"])
PY

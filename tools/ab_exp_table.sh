#!/bin/bash
# (round 6) Same-box A/B of the numerics contracts: v8's library (degree-13 Taylor exponential; build_ab/libcssm_pf_v8.so, built from the commit
# before the change) against the current one (v9: 64-entry table + degree-6 polynomial), interleaved, the bench's figures of both.
mkdir -p gpurun_out
for i in 1 2 3; do
  CSSM_PF_LIB=$PWD/build_ab/libcssm_pf_v8.so python3 bench.py --steps 500 --warmup 20 --no-cpu --no-pmc --no-generic > gpurun_out/abe_v8_$i.json 2>/dev/null || exit 1
  python3 bench.py --steps 500 --warmup 20 --no-cpu --no-pmc --no-generic > gpurun_out/abe_v9_$i.json 2>/dev/null || exit 1
done
for i in 1 2; do
  CSSM_PF_LIB=$PWD/build_ab/libcssm_pf_v8.so python3 bench.py --model c4 --steps 100 --warmup 10 --no-cpu > gpurun_out/abe4_v8_$i.json 2>/dev/null || exit 1
  python3 bench.py --model c4 --steps 100 --warmup 10 --no-cpu > gpurun_out/abe4_v9_$i.json 2>/dev/null || exit 1
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abe_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline_16m"]
    print(f, "us/step %.2f" % (j["ms_per_step"] * 1e3), "2^20:", {k: round(v, 2) for k, v in j["kernels_us"].items()},
          "2^24 d3:", {k: round(v, 1) for k, v in r["c2_d3"]["kernels_us"].items()}, "step %.1f" % r["c2_d3"]["step_us"], "d1:", {k: round(v, 1) for k, v in r["c1_d1"]["kernels_us"].items()}, "step %.1f" % r["c1_d1"]["step_us"],
          "d1 block tiles:", {k: round(v, 1) for k, v in r["c1_d1_block_tiles"]["kernels_us"].items()})
for f in sorted(glob.glob("gpurun_out/abe4_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "LGCP 2^24 us/event %.1f" % (j["ms_per_step"] * 1e3), {k: round(v, 1) for k, v in j["kernels_us"].items()})
PY

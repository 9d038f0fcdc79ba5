for i in 1 2; do
  for v in "CSSM_WAVE_SUMS=0" "CSSM_WAVE_SUMS=1" "CSSM_WAVE_SUMS=1 CSSM_NO_OFFW=1"; do
    env $v python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-generic > gpurun_out/abq.json 2>/dev/null || exit 1
    python3 - "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/abq.json").read().strip().splitlines()[-1]); r = j["roofline_16m"]
print(sys.argv[1], "2^24 d3:", {k: round(v, 1) for k, v in r["c2_d3"]["kernels_us"].items()}, "d1:", {k: round(v, 1) for k, v in r["c1_d1"]["kernels_us"].items()})
PY
  done
done

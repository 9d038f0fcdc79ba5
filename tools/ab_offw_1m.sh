for i in 1 2 3; do for v in 0 2; do
  CSSM_WAVE_SUMS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --no-generic --no-16m > gpurun_out/abo.json 2>/dev/null || exit 1
  python3 - "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/abo.json").read().strip().splitlines()[-1])
print("CSSM_WAVE_SUMS=" + sys.argv[1], "2^20: step %.2f" % (j["device_loop_ms"] * 1e3 / j["steps"]), {k: round(v, 2) for k, v in j["kernels_us"].items()})
PY
done; done

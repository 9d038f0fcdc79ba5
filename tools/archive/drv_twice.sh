for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv_$i.json 2> gpurun_out/drv_$i.err || exit 1; done
python3 - <<'PY'
import json, glob
f3 = lambda xs: " ".join("%.3f" % x for x in xs)
for f in sorted(glob.glob("gpurun_out/drv_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "us/step %.2f" % (j["ms_per_step"] * 1e3), "loop %.4f" % j["device_loop_ms"], j["kernels_us"])
    print("   wall", f3(j["wall_ms_each"])); print("   call", f3(j["call_ms_each"])); print("   dev ", f3(j["device_ms_each"]))
PY

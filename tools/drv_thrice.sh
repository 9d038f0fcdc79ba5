for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/fin20_$i.json 2> gpurun_out/fin20_$i.err || exit 1; done
python3 - <<'PY'
import json, glob
f3 = lambda xs: " ".join("%.3f" % x for x in xs)
for f in sorted(glob.glob("gpurun_out/fin20_?.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value %.3e us/step %.2f loop %.2f host_share %.3f" % (j["value"], j["ms_per_step"] * 1e3, j["device_loop_ms"] * 1e3 / 20, j["host_share"]))
    print("   wall", f3(j["wall_ms_each"])); print("   dev ", f3(j["device_ms_each"]))
PY

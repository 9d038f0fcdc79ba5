import os, sys, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(3):
    for v in ("1", "0"):
        env = dict(os.environ); env["HSA_ENABLE_INTERRUPT"] = v
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu", "--no-16m", "--no-pmc", "--no-generic"], env=env, capture_output=True, text=True).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        print(f"HSA_ENABLE_INTERRUPT={v}: {d['ms_per_step']*1e3:.2f} us/step, legs {[round(w*1e3) for w in d['wall_ms_each']]} us, device loop {d['device_loop_ms']*1e3:.0f} us", flush=True)

import json,sys
j=json.load(open(sys.argv[1]))
print("value %.4g  us/step %.2f  device_loop us/step %.2f  kern %s  frac %.3f" % (j["value"], j["ms_per_step"]*1e3, j["device_loop_ms"]*1e3/j["steps"], {k:round(v,2) for k,v in j["kernels_us"].items()}, j["roofline"]["frac"]))
for k,v in j.get("roofline_16m",{}).items(): print(" ", k, round(v["frac"],3), round(v["avg_launch_us"],1), {a:round(b,1) for a,b in v["kernels_us"].items()}, "step", round(v["step_us"],1))

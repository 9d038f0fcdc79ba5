"""One line per rank of a bench.py --gpus N (or --sharded) JSON line: usage shardsum.py file.json"""
import json, sys
j = json.load(open(sys.argv[1]))
print("value %.4g  us/step %.2f  n_gpus %d  legs(ms) %s" % (j["value"], j["ms_per_step"] * 1e3, j["n_gpus"], [round(x, 2) for x in j["wall_ms_each"]]))
for p in j["per_rank"]:
    print("  rank", p["rank"], p.get("kernels_us"), "collective", p.get("collective_us"), "x", p.get("collectives_per_observation"), p["legs"][-1])
    print("       legs wall/device ms:", [(l.get("wall_ms"), l.get("device_ms"), l.get("level_redos")) for l in p["legs"]])

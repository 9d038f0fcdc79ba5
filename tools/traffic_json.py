#!/usr/bin/env python3
"""profiles/traffic_latest.json from a tools/pmc_run.sh summary.

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB).  The factor 2 on FETCH_SIZE
is the gfx950 correction of MI355X_MICROARCH.md (HBM section), and it is calibrated for this access
pattern: at N = 2^24 (cloud 16x larger than the Infinity Cache) k_propagate<3> reports FETCH_SIZE =
230.2 MB against 469.8 MB of algorithmic reads (ratio 2.04), WRITE_SIZE = 537 MB = algorithmic writes.
"""
import json
import sys

src, dst, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
s = json.load(open(src))
out = {"particles": n, "unit": "bytes per launch", "correction": "2*FETCH_SIZE(KiB)*1024 + WRITE_SIZE(KiB)*1024"}
for k, v in s.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        name = k.split("<")[0]
        out[name + "_bytes_per_launch"] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
        out[name + "_fetch_kib_raw"] = v["FETCH_SIZE"]
        out[name + "_write_kib"] = v["WRITE_SIZE"]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))

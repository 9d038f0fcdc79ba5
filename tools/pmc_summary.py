#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/pmc_run.sh: per-kernel mean counter value per launch."""
import collections
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, f"pmc_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {}
for k, cs in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    out[k] = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
    out[k]["launches"] = max(v[1] for v in cs.values())
    line = f"{k:34s}"
    for c, v in sorted(out[k].items()):
        line += f" {c}={v:.4g}"
    print(line)
json.dump(out, open(os.path.join(root, f"pmc_{tag}_summary.json"), "w"), indent=1)

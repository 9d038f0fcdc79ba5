#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy per kernel of libcssm_pf (hipcc remarks)."""
import re
import subprocess
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# usage: kernel_resources.py [name-filter] [source.hip] [extra hipcc flags, e.g. -DCSSM_PROP_D=3]
SRC = os.path.join(HERE, "..", "composablestatespacemodels_amd", "csrc", sys.argv[2] if len(sys.argv) > 2 else "cssm_pf.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-mfma", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for line in out.splitlines():
    m = re.search(r"remark: ([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.groups()
    if k.strip() == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.split("(")[0].strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k.strip()] = v
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for name, r in rows.items():
    if pat in name:
        print(f"{name:40s} VGPR {r.get('VGPRs','?'):>4} SGPR {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize','?'):>5} "
              f"LDS {r.get('LDS Size','?'):>6} occ {r.get('Occupancy','?')}")

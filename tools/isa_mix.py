#!/usr/bin/env python3
"""Static instruction mix of one kernel of libcssm_pf: tools/isa_mix.py 'k_propagate<3, false>'."""
import collections
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "composablestatespacemodels_amd", "csrc", os.environ.get("ISA_SRC", "cssm_pf.hip"))
OUT = "/tmp/cssm_isa"
os.makedirs(OUT, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-mfma", "--offload-arch=gfx950",
                "-save-temps=obj", *sys.argv[4:], "-c", SRC, "-o", os.path.join(OUT, "x.o")], capture_output=True)
s = open(os.path.join(OUT, os.path.splitext(os.environ.get("ISA_SRC", "cssm_pf.hip"))[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
want = sys.argv[1] if len(sys.argv) > 1 else "k_propagate<3, false>"
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if want not in name:
        continue
    c = collections.Counter()
    for line in m.group(2).splitlines():
        line = line.strip()
        if not line or line[0] in ".;" or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    print(name.split("(")[0], "static instructions:", sum(c.values()))
    groups = collections.Counter()
    for op, n in c.items():
        if op.endswith("_f64") or "f64" in op: groups["fp64 " + ("div-ish" if "div" in op or "rcp" in op else "")] += n
        elif op.startswith("v_mad_u64") or op.startswith("v_mul_hi") or op.startswith("v_mul_lo"): groups["int mul"] += n
        elif op.startswith("v_"): groups["other VALU"] += n
        elif op.startswith("s_"): groups["SALU/branch"] += n
        else: groups["mem"] += n
    print(dict(groups))
    for op, n in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        print(f"  {op:28s}{n}")
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(m.group(2))
    break

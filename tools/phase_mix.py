#!/usr/bin/env python3
"""Static instruction mix of the fused kernel's phases (tools/phase_mix.hip: each phase a kernel of its own) -> a table for profiles/.
usage (build container): python tools/phase_mix.py > profiles/r06_isa_mix_d1.txt"""
import collections, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/phase_mix.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-mfma", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                       "-I", os.path.join(R, "include"), "-I", os.path.join(R, "composablestatespacemodels_amd", "csrc"),
                       os.path.join(R, "tools", "phase_mix.hip"), "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
def classify(op):
    if not op.startswith("v_"): return None
    if "f64" in op: return "fp64"
    if op.startswith(("v_mad_u64", "v_mul_hi", "v_mul_lo", "v_mad_u32")): return "int mul"
    if "dpp" in op: return "dpp"
    return "int / move"
rows = {}
for m in re.finditer(r"^(ph_\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
    c = collections.Counter()
    for line in m.group(2).splitlines():
        line = line.strip()
        if line and line[0] not in ".;" and not line.endswith(":"):
            k = classify(line.split()[0])
            if k: c[k] += 1
    rows[m.group(1)] = c
base = rows["ph_base"]
print("phase                  VALU   fp64  int-mul  int/move  dpp   per particle-step at d = 1")
share = {"ph_philox": 0.25, "ph_boxmuller": 0.5, "ph_exp": 1, "ph_exp_le0": 1, "ph_fix": 0.5, "ph_wavesum": 1 / 16.0}
tot = 0.0
for k in ("ph_philox", "ph_boxmuller", "ph_exp", "ph_exp_le0", "ph_fix", "ph_wavesum"):
    c = rows[k] - base if False else rows[k]
    n = sum(c.values()) - sum(base.values())
    per = n * share[k]
    tot += per
    print(f"{k[3:]:20s} {n:6d} {c['fp64']:6d} {c['int mul']:8d} {c['int / move'] - base['int / move']:9d} {c['dpp']:5d}   x {share[k]:<6g} = {per:6.1f}")
print(f"sum of the phases above: {tot:.1f} VALU instructions per particle-step")

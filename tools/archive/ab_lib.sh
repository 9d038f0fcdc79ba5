#!/bin/bash
# A/B of two BUILDS of the library on one box (boxes of the pool differ by +-5 %): alternating processes, per-kernel event times.
# usage (GPU box, repo root): tools/archive/ab_lib.sh <libA.so> <libB.so> [c2|c1] [sizes, comma separated] [rounds]
# The variant builds are made beforehand, e.g.  cp libcssm_pf.so build_ab/A.so  before a change and  cp ... build_ab/B.so  after it.
A=$1; B=$2; M=${3:-c2}; S=${4:-1048576}; R=${5:-3}
for i in $(seq 1 $R); do
  CSSM_PF_LIB=$A python3 tools/archive/kernel_probe.py $M $S 1 || exit 1
  CSSM_PF_LIB=$B python3 tools/archive/kernel_probe.py $M $S 1 || exit 1
done

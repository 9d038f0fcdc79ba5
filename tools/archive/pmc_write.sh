#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the kernels of a short bench run, for an experiment build: tools/pmc_write.sh <tag> [lib]
TAG=$1; LIB=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
[ -n "$LIB" ] && export CSSM_PF_LIB=$R/composablestatespacemodels_amd/csrc/$LIB
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_${TAG}_$c -- python3 $R/bench.py --steps 60 --warmup 2 --no-cpu --no-16m --no-pmc --repeats 1 > $R/gpurun_out/pmcw_${TAG}_$c.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"gpurun_out/pmcw_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), v in sorted(acc.items()):
    if k.startswith("k_") and "diag" not in k and "init" not in k:
        print(f"{tag} {k:40s} {c} = {v[0] / v[1] / 1024:.2f} MiB per launch ({v[1]} launches)")
PY

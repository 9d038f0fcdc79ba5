#!/bin/bash
# PMC counters of the resampling kernel, old (k_offspring_self) against new (k_offspring_wave), same box: tools/pmc_ab_offspring.sh <particles>
N=${1:-1048576}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for v in 0 1; do
  for pass in "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    set -- $pass; name=$1; shift
    CSSM_WAVE_SUMS=$v rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ab${v}_${N}_${name} -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu --no-16m --no-pmc --no-generic --repeats 1 --particles $N > $R/gpurun_out/pmc_ab${v}_${N}_${name}.log 2>&1
  done
done
cd $R
python3 - $N <<'PY'
import csv, glob, collections, sys
N = int(sys.argv[1])
for v in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(f"gpurun_out/pmc_ab{v}_{N}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "offspring" in k:
                a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, cs in acc.items():
        m = {c: x[0] / x[1] for c, x in cs.items()}
        print(k, "VALU/particle %.1f SALU %.1f LDS %.1f VMEM_RD %.2f VMEM_WR %.2f waves %d wait/wavecyc %.2f busy_cycles %.0f wave_cycles %.3g active_valu %.3g" % (
            m["SQ_INSTS_VALU"] * 64 / N, m["SQ_INSTS_SALU"] * 64 / N, m["SQ_INSTS_LDS"] * 64 / N, m["SQ_INSTS_VMEM_RD"] * 64 / N, m["SQ_INSTS_VMEM_WR"] * 64 / N,
            m["SQ_WAVES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_BUSY_CYCLES"], m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"]))
PY

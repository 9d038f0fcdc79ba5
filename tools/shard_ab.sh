for rep in 1 2; do
for lib in libcssm_pf.so libcssm_pf_noasm.so; do
python - <<PY
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import composablestatespacemodels_amd._abi as abi
abi.LIB_PATH = os.path.join("composablestatespacemodels_amd", "csrc", "$lib")
os.environ["PROBE_FIXED_ONLY"] = "1"
sys.argv = ["sharded_probe.py", "1048576"]
exec(open("tools/sharded_probe.py").read().replace('os.environ["MASTER_PORT"] = "29544"', 'os.environ["MASTER_PORT"] = str(29544 + $rep * 7 + len("$lib"))'))
PY
done; done 2>&1 | grep "sharded RCCL" | sed 's/^/  /'

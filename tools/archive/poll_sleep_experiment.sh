set -e
cd composablestatespacemodels_amd/csrc && mkdir -p build_stamps
for SL in 2 8 32 127; do
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mfma --offload-arch=gfx950 -Wno-unused-function -DCSSM_OFF_STAMPS -DCSSM_POLL_LL_SLEEP=$SL -c -o build_stamps/shard.o cssm_shard.hip
  hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_stamps/libcssm_pf_stamps.so build/pf.o build_stamps/shard.o build/batch.o build/residual.o build/model.o build/rtc.o build/prop_d*.o -ldl
  echo "== s_sleep $SL"
  (cd ../.. && CSSM_PF_LIB=$PWD/composablestatespacemodels_amd/csrc/build_stamps/libcssm_pf_stamps.so python tools/archive/exchange_stamps.py 2>&1 | grep -v amdgpu | grep "pack:\|all flags\|headers in LDS\|own ancestors\|rows expanded")
done

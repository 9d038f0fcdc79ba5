set -e
cd composablestatespacemodels_amd/csrc && mkdir -p build_stamps
for V in "" "-DCSSM_EXP_NO_PRIO"; do   # (the macro is gone: the priority is in the code; kept as the record of the A/B)
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mfma --offload-arch=gfx950 -Wno-unused-function -DCSSM_OFF_STAMPS $V -c -o build_stamps/shard.o cssm_shard.hip
  hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o build_stamps/libcssm_pf_stamps.so build/pf.o build_stamps/shard.o build/batch.o build/residual.o build/model.o build/rtc.o build/prop_d*.o -ldl
  echo "== variant [$V]"
  (cd ../.. && for i in 1 2 3; do CSSM_PF_LIB=$PWD/composablestatespacemodels_amd/csrc/build_stamps/libcssm_pf_stamps.so python tools/archive/exchange_stamps.py 2>&1 | grep "header + flag\|all flags\|own ancestors" | tr '\n' ' '; echo; done)
done

// cssm_series.hip -- the k_series instantiations of ONE latent dimension (compile with -DCSSM_SER_D=<d>).
#include "cssm_host.h"
#include "cssm_series.hip.h"

#ifndef CSSM_SER_D
#error "compile with -DCSSM_SER_D=<latent dimension>"
#endif

#define CSSM_CAT2(a, b) a##b
#define CSSM_CAT(a, b) CSSM_CAT2(a, b)

#if CSSM_SER_D == 1
size_t cssm_series_sync_bytes() { return sizeof(SeriesSync); }
#endif

static const void* series_kernel(int obs) {
  constexpr int D = CSSM_SER_D;
  if (obs == CSSM_OBS_POISSON) return (const void*)k_series<D, CSSM_OBS_POISSON>;
  if (obs == CSSM_OBS_GAUSSIAN) return (const void*)k_series<D, CSSM_OBS_GAUSSIAN>;
  return (const void*)k_series<D, -1>;
}

hipError_t CSSM_CAT(cssm_series_occupancy_d, CSSM_SER_D)(int obs, size_t smem, int* blocks_per_cu) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, series_kernel(obs), CSSM_BLOCK, smem);
}

hipError_t CSSM_CAT(cssm_series_launch_d, CSSM_SER_D)(const SeriesLaunch& a) {
  SeriesSync* sy = (SeriesSync*)a.sync;
  uint64_t n = a.n, seed = a.seed;
  void* args[] = {(void*)&a.state0, (void*)&a.state1, (void*)&a.stride, (void*)&a.anc, (void*)&a.logw, (void*)&n, (void*)&seed,
                  (void*)&a.recs, (void*)&a.T, (void*)&a.mk, (void*)&a.sc, (void*)&sy, (void*)&a.logtab, (void*)&a.per_block,
                  (void*)&a.cur0, (void*)&a.force_exact, (void*)&a.ll_t, (void*)&a.ess_t, (void*)&a.path, (void*)&a.ts, (void*)&a.ts_blocks};
  // cooperative: the runtime guarantees that all blocks are resident together (they spin on each other)
  return hipLaunchCooperativeKernel(series_kernel(a.obs), dim3(a.grid), dim3(CSSM_BLOCK), args, (unsigned)a.smem, a.stream);
}

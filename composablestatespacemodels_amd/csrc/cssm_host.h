// cssm_host.h -- what the translation units of libcssm_pf share on the host side (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cssm_device.hip.h"

// Everything one k_propagate launch needs.  cssm_pf.hip fills it, the per-dimension translation unit that owns the
// instantiation (cssm_prop.hip compiled with -DCSSM_PROP_D=<d>) launches it.
struct PropLaunch {
  int grid;
  hipStream_t stream;
  int lgcp;            // k_propagate<D, true, ...>
  int obs;             // CSSM_OBS_POISSON / CSSM_OBS_GAUSSIAN compiled in, -1 = runtime observation kind
  int sums;            // the kernel also forms the fixed-point sums of exp(w - c)
  const double* src; size_t src_stride; const uint32_t* anc;
  double* dst; size_t dst_stride; double* logw; uint64_t n; uint64_t gid0; uint64_t seed;
  const StepRec* rec; ModelK mk; Scalars* sc; int slot_set;
  const double* src2; size_t src2_stride; uint32_t n_split; const double* logtab;
  uint64_t chunk; int do_sums; cssm_u128* subS; cssm_u128* subS2; double* pick_out; uint32_t pick_slot;
  const double* fsub;  // LGCP with a time-dependent f: the sub-step coefficient table (else nullptr)
  int shard_slim;      // the sharded filter's slim launch k_propagate_shard applies (cssm_pf.hip decides)
  int one;             // k_propagate_self<..., ONE>: 1 = the block's range is one tile, 2 = the same body tile after tile, 0 = software-pipelined
};

// one per latent dimension, defined in cssm_prop.hip
#define CSSM_DECL_PROP(D) void cssm_prop_launch_d##D(const PropLaunch& a);
CSSM_DECL_PROP(1) CSSM_DECL_PROP(2) CSSM_DECL_PROP(3) CSSM_DECL_PROP(4) CSSM_DECL_PROP(5) CSSM_DECL_PROP(6) CSSM_DECL_PROP(7) CSSM_DECL_PROP(8)
CSSM_DECL_PROP(9) CSSM_DECL_PROP(10) CSSM_DECL_PROP(11) CSSM_DECL_PROP(12) CSSM_DECL_PROP(13) CSSM_DECL_PROP(14) CSSM_DECL_PROP(15) CSSM_DECL_PROP(16)
#undef CSSM_DECL_PROP

// ---- k_step (cssm_propagate.hip.h): resampling of observation s - 1 + propagate of observation s in one launch
struct StepLaunch {
  int grid; hipStream_t stream; int obs; uint32_t chunk;   // chunk: particles per block = per unit of sums (CSSM_TILE or half)
  const double* src; size_t src_stride; double* dst; size_t dst_stride; const double* logw_in; double* logw_out; uint32_t n; uint64_t seed;
  const StepRec* rec; ModelK mk; Scalars* sc; int set_in; const double* logtab;
  const cssm_u128* inS; const cssm_u128* inS2; cssm_u128* outS; cssm_u128* outS2; uint32_t nunits;
  double* ll_t; int32_t* ess_t; int force_exact; double* pick_out; uint32_t pick_slot;
};
#define CSSM_DECL_STEP(D) void cssm_step_launch_d##D(const StepLaunch& a);
CSSM_DECL_STEP(1) CSSM_DECL_STEP(2) CSSM_DECL_STEP(3) CSSM_DECL_STEP(4) CSSM_DECL_STEP(5) CSSM_DECL_STEP(6) CSSM_DECL_STEP(7) CSSM_DECL_STEP(8)
CSSM_DECL_STEP(9) CSSM_DECL_STEP(10) CSSM_DECL_STEP(11) CSSM_DECL_STEP(12) CSSM_DECL_STEP(13) CSSM_DECL_STEP(14) CSSM_DECL_STEP(15) CSSM_DECL_STEP(16)
#undef CSSM_DECL_STEP

// ---- the persistent series kernel (cssm_series.hip.h), one translation unit per latent dimension (cssm_series.hip)
struct SeriesLaunch {
  int grid; hipStream_t stream; int obs; size_t smem;
  double* state0; double* state1; size_t stride; uint32_t* anc; double* logw; uint64_t n; uint64_t seed;
  const StepRec* recs; uint32_t T; ModelK mk; Scalars* sc; void* sync; const double* logtab; uint32_t per_block; int cur0;
  int force_exact; double* ll_t; int32_t* ess_t; double* path; unsigned long long* ts; uint32_t ts_blocks;
};
#define CSSM_DECL_SER(D) hipError_t cssm_series_launch_d##D(const SeriesLaunch& a); \
                         hipError_t cssm_series_occupancy_d##D(int obs, size_t smem, int* blocks_per_cu);
CSSM_DECL_SER(1) CSSM_DECL_SER(2) CSSM_DECL_SER(3) CSSM_DECL_SER(4) CSSM_DECL_SER(5) CSSM_DECL_SER(6) CSSM_DECL_SER(7) CSSM_DECL_SER(8)
CSSM_DECL_SER(9) CSSM_DECL_SER(10) CSSM_DECL_SER(11) CSSM_DECL_SER(12) CSSM_DECL_SER(13) CSSM_DECL_SER(14) CSSM_DECL_SER(15) CSSM_DECL_SER(16)
#undef CSSM_DECL_SER
size_t cssm_series_sync_bytes();   // sizeof(SeriesSync)

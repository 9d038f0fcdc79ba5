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
  int sharded;         // a shard's launch: both sums (S and S2) travel in the exchange -- never the single-GPU _self kernels, which
                       //   leave the sum of squares to k_offspring
  int shard_slim;      // the sharded filter's slim launch k_propagate_shard applies (cssm_pf.hip decides)
  uint32_t step;       // the observation's index (= rec->step, read from the host copy of the record: k_propagate_self / _shard use it before any load lands)
  int one;             // k_propagate_self<..., ONE>: 1 = the block's range is one tile, 2 = the same body tile after tile, 0 = software-pipelined
  int specialise;      // 1 (default): the kernel that holds the model's structure at compile time -- an ahead-of-time instantiation where one exists
                       //   (cssm_prop.hip: KnownStructures), else one compiled at run time (cssm_rtc.cpp); 0: the structure-as-data kernel
  const void* chains;  // batched launch (cssm_batch.hip): the device array of ChainBase, nchains of them (grid.y); src / dst / anc / logw / sc /
  int nchains;         //   rec / seed / subS / pick_* above are unused -- the chains' own come from the array
  int cur, anc_valid, want_pick; uint32_t rec_idx;
  int obs_kind;        // the handle's CSSM_OBS_* (a run-time-compiled kernel holds it at compile time whichever it is; `obs` above is -1 for
                       //   the densities the ahead-of-time kernels keep behind a switch)
};

// cssm_rtc.cpp: the fused kernel of launch `a` specialised at run time (kind 0 = k_propagate_self, 1 = k_propagate_shard, 2 = the LGCP
// k_propagate); false = not launched, the caller falls back to the structure-as-data kernel
bool cssm_rtc_launch(const PropLaunch& a, int kind, int D, int IT, int onev);

// one per latent dimension, defined in cssm_prop.hip; returns what the kernel it chose does beyond the propagate itself
#define CSSM_PROP_LAUNCHED_GRP 1   /* its blocks accumulate the sums of groups of units (Scalars::grp; asked for by bit 8 of slot_set) */
#define CSSM_PROP_LAUNCHED_WS 2    /* its waves own contiguous quarter units and stored their exact sums (subS2, four per block; asked for by bit 13 of slot_set) */
#define CSSM_DECL_PROP(D) int cssm_prop_launch_d##D(const PropLaunch& a);
CSSM_DECL_PROP(1) CSSM_DECL_PROP(2) CSSM_DECL_PROP(3) CSSM_DECL_PROP(4) CSSM_DECL_PROP(5) CSSM_DECL_PROP(6) CSSM_DECL_PROP(7) CSSM_DECL_PROP(8)
CSSM_DECL_PROP(9) CSSM_DECL_PROP(10) CSSM_DECL_PROP(11) CSSM_DECL_PROP(12) CSSM_DECL_PROP(13) CSSM_DECL_PROP(14) CSSM_DECL_PROP(15) CSSM_DECL_PROP(16)
#undef CSSM_DECL_PROP

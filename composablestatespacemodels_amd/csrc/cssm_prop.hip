// cssm_prop.hip -- the k_propagate instantiations of ONE latent dimension (compile with -DCSSM_PROP_D=<d>); the
// Makefile builds the 16 objects in parallel.
#include "cssm_host.h"
#include "cssm_propagate.hip.h"

#ifndef CSSM_PROP_D
#error "compile with -DCSSM_PROP_D=<latent dimension>"
#endif

#ifndef CSSM_PROP_SELF
#define CSSM_PROP_SELF 1   /* 0: always the generic kernel (experiments) */
#endif
#define CSSM_CAT2(a, b) a##b
#define CSSM_CAT(a, b) CSSM_CAT2(a, b)

// Structures with instantiations of their own (k_propagate_self / k_propagate_shard<..., MKW, MKW1, MKW2>): ModelK::comp[0..2] of the
// reference's example models -- byte per component = SDE kind | FM_* << 2 | leaf ends << 4 | leftmost leaf << 5 (cssm_model.cpp).
// Poisson observation (d = 1: also Gaussian -- Model.linear), fused sums (the batch drivers' default kernels).
//   d = 1: one leaf on Brownian motion (configs[0]), generalised Brownian motion, an OU process
//   d = 3: poisson(ou(1)) |+| seasonal(24, 1, ou(2))                      (configs[1], the bench model)
//   d = 9: poisson(brownianMotion(1)) |+| seasonal(24, 4, ouProcess(8))   (configs[2])
template <int D> struct KnownStructures { static constexpr int n = 0; static constexpr uint32_t w[1][3] = {{0u, 0u, 0u}}; };
template <> struct KnownStructures<1> { static constexpr int n = 3; static constexpr uint32_t w[3][3] = {{0x34u, 0u, 0u}, {0x35u, 0u, 0u}, {0x36u, 0u, 0u}}; };
template <> struct KnownStructures<3> { static constexpr int n = 1; static constexpr uint32_t w[1][3] = {{0x001A0636u, 0u, 0u}}; };
template <> struct KnownStructures<9> { static constexpr int n = 1; static constexpr uint32_t w[1][3] = {{0x0A0A0634u, 0x0A0A0A0Au, 0x0000001Au}}; };

// the fused-sums Poisson kernel of a known structure (ONEV as in OneTile; 0: the pipelined kernel; SHARD: the sharded slim
// launch): true if one was launched
template <int D, int IT, int ONEV, bool SHARD, int OB, int I> struct TryKnown {
  static bool go(const PropLaunch& a) {
    if constexpr (I >= KnownStructures<D>::n) {
      return false;
    } else {
      constexpr uint32_t W0 = KnownStructures<D>::w[I][0], W1 = KnownStructures<D>::w[I][1], W2 = KnownStructures<D>::w[I][2];
      if (a.mk.comp[0] == W0 && a.mk.comp[1] == W1 && a.mk.comp[2] == W2) {
        if constexpr (SHARD)
          k_propagate_shard<D, IT, OB, ONEV, W0, W1, W2><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(
              a.src, a.src_stride, a.anc, a.dst, a.dst_stride, a.logw, a.n, a.gid0, a.seed, a.rec, a.mk, a.sc, a.src2, a.n_split, a.logtab, a.chunk,
              a.subS, a.subS2, a.step, a.slot_set);
        else
          k_propagate_self<D, IT, OB, 1, ONEV, W0, W1, W2><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(
              a.src, a.src_stride, a.anc, a.dst, a.dst_stride, a.logw, a.n, a.seed, a.rec, a.mk, a.sc, a.slot_set, a.logtab, a.chunk, a.subS, a.subS2,
              a.pick_out, a.pick_slot, a.step);
        return true;
      }
      return TryKnown<D, IT, ONEV, SHARD, OB, I + 1>::go(a);
    }
  }
};
// ... or, for every other structure (and every observation model), the instantiation compiled at run time (cssm_rtc.cpp)
template <int D, int IT, int ONEV, bool SHARD = false> static bool launch_known(const PropLaunch& a) {
  if (!a.specialise || D > 12 || a.mk.d != D) return false;
  // (specialise == 2, a verification setting: the run-time-compiled kernel also where an ahead-of-time instantiation exists)
  if (a.specialise != 2 && a.obs == CSSM_OBS_POISSON && TryKnown<D, IT, ONEV, SHARD, CSSM_OBS_POISSON, 0>::go(a)) return true;
  if constexpr (D == 1) { if (a.specialise != 2 && a.obs == CSSM_OBS_GAUSSIAN && TryKnown<D, IT, ONEV, SHARD, CSSM_OBS_GAUSSIAN, 0>::go(a)) return true; }   // (Model.linear on one component)
  return cssm_rtc_launch(a, SHARD ? 1 : 0, D, IT, ONEV);
}

// the single-tile instantiation of the small clouds (one pair per thread for d <= 8, one particle for d >= 9)
template <int D, int IT> struct OneTile {
  static void go(const PropLaunch& a) {
#define PROP_ONE(OB, ONEV)                                                                                                        \
  k_propagate_self<D, IT, OB, 1, ONEV><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(a.src, a.src_stride, a.anc, a.dst, a.dst_stride, \
      a.logw, a.n, a.seed, a.rec, a.mk, a.sc, a.slot_set, a.logtab, a.chunk, a.subS, a.subS2, a.pick_out, a.pick_slot, a.step)
    if (a.one == 1 ? launch_known<D, IT, 1>(a) : launch_known<D, IT, 2>(a)) return;
    if (a.one == 1) {          // the block's range is one tile
      if (a.obs == CSSM_OBS_POISSON) PROP_ONE(CSSM_OBS_POISSON, 1);
      else if (a.obs == CSSM_OBS_GAUSSIAN) PROP_ONE(CSSM_OBS_GAUSSIAN, 1);
      else PROP_ONE(-1, 1);
    } else {                   // tile after tile
      if (a.obs == CSSM_OBS_POISSON) PROP_ONE(CSSM_OBS_POISSON, 2);
      else if (a.obs == CSSM_OBS_GAUSSIAN) PROP_ONE(CSSM_OBS_GAUSSIAN, 2);
      else PROP_ONE(-1, 2);
    }
#undef PROP_ONE
  }
};

int CSSM_CAT(cssm_prop_launch_d, CSSM_PROP_D)(const PropLaunch& a) {
  constexpr int D = CSSM_PROP_D;
  constexpr int IT = PropItems<D>::value;
  if (a.chains != nullptr) {
    // B independent clouds (cssm_batch.hip): the fused-sums single-GPU kernels only; the model's structure and observation model at
    // compile time through the run-time compiler, else the structure-as-data instantiation with the observation model behind its switch
    const int onev = a.one;
    if (a.specialise && D <= 12 && cssm_rtc_launch(a, 3, D, IT, onev)) return (a.slot_set & 0x100) ? CSSM_PROP_LAUNCHED_GRP : 0;
#define PROP_BATCH(ONEV)                                                                                                               \
    k_propagate_batch<D, IT, -1, ONEV><<<dim3(a.grid, a.nchains), dim3(CSSM_BLOCK), 0, a.stream>>>(                                   \
        static_cast<const ChainBase*>(a.chains), a.cur, a.anc_valid, a.src_stride, a.n, a.rec_idx, a.mk, a.slot_set, a.logtab, a.chunk, a.want_pick, a.step)
    if (onev == 1) PROP_BATCH(1); else if (onev == 2) PROP_BATCH(2); else PROP_BATCH(0);
#undef PROP_BATCH
    return (a.slot_set & 0x100) ? CSSM_PROP_LAUNCHED_GRP : 0;
  }
#define PROP_GO(LG, OB, SM)                                                                                               \
  k_propagate<D, LG, IT, OB, SM><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(                                        \
      a.src, a.src_stride, a.anc, a.dst, a.dst_stride, a.logw, a.n, a.gid0, a.seed, a.rec, a.mk, a.sc, a.slot_set, a.src2, \
      a.src2_stride, a.n_split, a.logtab, a.chunk, a.do_sums, a.subS, a.subS2, a.pick_out, a.pick_slot, a.fsub)
  // the single-GPU lean launch: nothing of the sharded filter, no fused sums, no pick, first global id 0
  const bool self = (!a.sharded || !a.sums) && !a.lgcp && a.src2 == nullptr && a.gid0 == 0 && a.fsub == nullptr && (a.sums || a.pick_out == nullptr) && (!a.sums || a.do_sums);
#define PROP_SELF(OB, SM)                                                                                                   \
  k_propagate_self<D, IT, OB, SM><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(a.src, a.src_stride, a.anc, a.dst, a.dst_stride, \
      a.logw, a.n, a.seed, a.rec, a.mk, a.sc, a.slot_set, a.logtab, a.chunk, a.subS, a.subS2, a.pick_out, a.pick_slot, a.step)
  // small clouds: one tile of the kernel per block (half a tile of 1024 for d <= 8, a quarter for d >= 9): the ONE instantiation
  // every kernel that forms the sums adds to the group sums where bit 8 of the set argument asks for them (one block per unit: cssm_pf.hip)
  const int grp = (a.sums && a.do_sums && (a.slot_set & 0x100)) ? CSSM_PROP_LAUNCHED_GRP : 0;
  // ... and the single GPU's tile-after-tile launch stores its waves' sums where bit 13 asks for them (propagate_block: WR)
  const int ws = (CSSM_PROP_WR != 0 && self && CSSM_PROP_SELF && a.sums && a.do_sums && a.one == 2 && (a.slot_set & 0x2000)) ? CSSM_PROP_LAUNCHED_WS : 0;
  if (self && CSSM_PROP_SELF && a.sums && a.one) {
    OneTile<D, IT>::go(a);
  } else if (self && CSSM_PROP_SELF) {
    if (a.sums && launch_known<D, IT, 0>(a)) { /* a known structure's own instantiation */ }
    else if (a.obs == CSSM_OBS_POISSON) { if (a.sums) PROP_SELF(CSSM_OBS_POISSON, 1); else PROP_SELF(CSSM_OBS_POISSON, 0); }
    else if (a.obs == CSSM_OBS_GAUSSIAN) { if (a.sums) PROP_SELF(CSSM_OBS_GAUSSIAN, 1); else PROP_SELF(CSSM_OBS_GAUSSIAN, 0); }
    else { if (a.sums) PROP_SELF(-1, 1); else PROP_SELF(-1, 0); }
  } else if (!a.lgcp && CSSM_PROP_SELF && a.sums && a.do_sums && a.shard_slim) {
    // the sharded filter (single-collective exchange): the slim launch; tile after tile while a unit has few tiles
#define PROP_SHARD(OB, ONEV)                                                                                               \
  k_propagate_shard<D, IT, OB, ONEV><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(a.src, a.src_stride, a.anc, a.dst, a.dst_stride, a.logw, \
      a.n, a.gid0, a.seed, a.rec, a.mk, a.sc, a.src2, a.n_split, a.logtab, a.chunk, a.subS, a.subS2, a.step, a.slot_set)
    if (a.one == 2 ? launch_known<D, IT, 2, true>(a) : launch_known<D, IT, 0, true>(a)) {
      /* a known structure's own instantiation */
    } else if (a.one == 2) {
      if (a.obs == CSSM_OBS_POISSON) PROP_SHARD(CSSM_OBS_POISSON, 2); else if (a.obs == CSSM_OBS_GAUSSIAN) PROP_SHARD(CSSM_OBS_GAUSSIAN, 2); else PROP_SHARD(-1, 2);
    } else {
      if (a.obs == CSSM_OBS_POISSON) PROP_SHARD(CSSM_OBS_POISSON, 0); else if (a.obs == CSSM_OBS_GAUSSIAN) PROP_SHARD(CSSM_OBS_GAUSSIAN, 0); else PROP_SHARD(-1, 0);
    }
#undef PROP_SHARD
  } else if (a.lgcp) {
    // the log-Gaussian Cox process on one OU component (configs[3]): its structure at compile time (the sub-step loop branches
    // on it once per sub-step and particle)
    // SM: 0 = log-weights stored (the level is the max: the first event of a series, a redone one, the multinomial resampler);
    // 1 / 2 = the sums formed relative to the PREDICTED level (contract v8) and the weights stored -- one GPU / a shard (both sums)
#define PROP_LGCP(SM)                                                                                                          \
    do {                                                                                                                       \
      if (a.specialise == 1 && D == 1 && a.mk.comp[0] == 0x36u)                                                                   \
        k_propagate<D, true, IT, -1, SM, (D == 1 ? 0x36u : 0u)><<<dim3(a.grid), dim3(CSSM_BLOCK), 0, a.stream>>>(              \
            a.src, a.src_stride, a.anc, a.dst, a.dst_stride, a.logw, a.n, a.gid0, a.seed, a.rec, a.mk, a.sc, a.slot_set, a.src2, \
            a.src2_stride, a.n_split, a.logtab, a.chunk, a.do_sums, a.subS, a.subS2, a.pick_out, a.pick_slot, a.fsub);         \
      else if (cssm_rtc_launch(a, 2, D, IT, 0)) { /* the structure of any other LGCP model, compiled at run time */ }          \
      else                                                                                                                     \
        PROP_GO(true, -1, SM);                                                                                                 \
    } while (0)
    if (!a.sums) PROP_LGCP(0); else if (a.sharded) PROP_LGCP(2); else PROP_LGCP(1);
#undef PROP_LGCP
  } else if (a.obs == CSSM_OBS_POISSON) {   // (the generic kernel with sums serves sharded handles only: both sums, SUMS = 2)
    if (a.sums) PROP_GO(false, CSSM_OBS_POISSON, 2); else PROP_GO(false, CSSM_OBS_POISSON, 0);
  } else if (a.obs == CSSM_OBS_GAUSSIAN) {
    if (a.sums) PROP_GO(false, CSSM_OBS_GAUSSIAN, 2); else PROP_GO(false, CSSM_OBS_GAUSSIAN, 0);
  } else {
    if (a.sums) PROP_GO(false, -1, 2); else PROP_GO(false, -1, 0);
  }
#undef PROP_GO
#undef PROP_SELF
  return grp | ws;
}

#if defined(CSSM_PROP_STAMPS)
// diagnostic build: the stamps the slim kernels' blocks of THIS dimension left
extern "C" int CSSM_CAT(cssm_prop_debug_stamps_d, CSSM_PROP_D)(unsigned long long* out, size_t nwords) {
  if (nwords > 8192 * 8) nwords = 8192 * 8;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prop_stamps), nwords * 8) == hipSuccess ? 0 : 1;
}
#endif

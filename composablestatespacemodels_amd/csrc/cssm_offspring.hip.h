// cssm_offspring.hip.h -- the resampling kernel of libcssm_pf (Resampling.scala:36-72: treeEcdf, findAllInTreeMap, systematicResampling;
// :78-96 for the other two grids): unit prefix, cumulative weights, end slots, ancestor runs, ll and ESS.  Included by cssm_kernels.hip.h
// (round 5 moved it out of that file).
//
// Map of the file:
//   block_sum_u128, offspring_exact_counts, TotExact      helpers of the rare exact path
//   tile_end_slots, end_slot_of_prefix                    a tile's end slots: the fp64 fast path and its flagged band
//   publish_observation                                   what the single-GPU launch's publisher block files
//   offspring_body                                        the order of everything for every launch that resamples, in five stretches:
//        1 entry: ONE round of loads (hold word, first tile's weights, the unit / group sums, the record's scalars, the max slots)
//        2 level: the max decoded (single GPU) or taken from the ranks' words (a shard); a ruled-out level ends the block
//        3 sums -> total + the block's prefix (`scan_units`: every unit sum / layout 1 / layout 2 of Scalars::grp), the publisher's
//          path, and -- a shard on group sums -- `mid`: the wait for the peers' headers, AFTER all local work
//        4 tile loop, front half: weights on the 2^-96 grid, thread sums, wave scan, block barrier (`tile_front`; EARLY: the first tile's
//          front half ran ahead of stretch 3's barrier)
//        5 tile loop, back half: exclusive prefixes -> tile_end_slots -> fill_runs_wave (cssm_device.hip.h), running prefix advanced
//   k_offspring, k_offspring_self                         the launches (cssm_shard_kernels.hip.h and cssm_batch.hip hold the others)
// offspring_body stays ONE function on purpose.  Round 5 took it apart into traits + five __forceinline__ functions (request_sums,
// scan_sums, publisher, tile_front, tile_back; the attempt is in the history of this file's first commit) and measured the register
// allocation of k_offspring_self after each extraction: the same code behind a function boundary kept a dozen uniform values live that
// the single function rematerialises -- scalar registers 93 -> 106, which then spill into vector lanes and from there to scratch:
// 16-28 bytes per thread in the default instantiations (tile_back), 84 in <0, 2, 0> (tile_front), 119 -> 128 VGPRs + 48 bytes in the
// stratified one (scan_sums).  This kernel was brought from 28 bytes of scratch to none in round 2 for 7 MB of write-back per launch;
// the stretches above are marked in the body instead.
#pragma once

#include "cssm_device.hip.h"

// ------------------------------------------------------------------------------------ offspring (end slots)

#define CSSM_RUN_DIRECT 8 /* runs up to this length are written by the owning thread */

// treeEcdf (model/Resampling.scala:52-58): C_j = (sum_{i<=j} w1_i) / (sum_i w1_i), here the
// correctly rounded quotient of the exact fixed-point sums; end slot of particle j =
// #{ i : (u+i)/N <= C_j } (the ks of :69 against the keys of the TreeMap).
// FUSE: also do findAllInTreeMap (:36-46): the ancestors of the slots a wave's 256 particles own -- the runs
// [end_{j-1}, end_j) <- j -- are assembled in the wave's LDS region and written as whole lines (fill_runs_wave), so on a
// single GPU the end slots never travel through HBM.  The exact exchange of the sharded filter has the end slots
// stored as well.  A block walks the tiles of one unit with a running prefix.
// SELF (single GPU): there is no scan kernel.  Every block sums the <= ~1K unit totals itself (integer
// sums: every block gets the same bits); one more block, the publisher, files max / totals / ll / ess and clears the
// other max-slot sets for the next weighted steps.
// RS = CSSM_RESAMPLE_* at compile time: the systematic kernel must not carry the stratified path's Philox code
// (it cost 40 VGPRs and a wave of occupancy when the kind was a runtime argument).
#ifndef CSSM_OFF_WAVES
#define CSSM_OFF_WAVES 4
#endif
// unit sums every block of k_offspring_self requests before anything else, per thread (x 256 threads); the rest in a loop
#ifndef CSSM_OFF_UPRE
#define CSSM_OFF_UPRE 4   /* 1024 entries (every cloud from 2^20 particles on) in flight at once; smaller clouds on half / quarter tiles: a loop for the rest */
#endif
// k_offspring's ancestor lines: 1 = write-through (sc1) stores, 0 = plain stores (dirty lines written back when the kernel ends)
#ifndef CSSM_OFF_SC1
#define CSSM_OFF_SC1 1
#endif
// sum over the block's threads, the same value in every thread (s_red: CSSM_BLOCK / 64 entries of LDS; all threads call)
__device__ __forceinline__ cssm_u128 block_sum_u128(cssm_u128 v, cssm_u128* s_red) {
  v = wave_sum_u128(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  cssm_u128 t = s_red[0];
#pragma unroll
  for (int w = 1; w < CSSM_BLOCK / 64; ++w) t = cssm_u128_add(t, s_red[w]);
  return t;
}

// The exact predicate of the systematic end slots for the single-GPU kernel: taken for about N * 2^-43 of the particles.  Inlined
// once per particle of a thread, inside the loop over them, it was 60 % of the kernel's code and what its registers were sized by
// (values spilled to scratch on the hot path, ahead of the branch; as a real call the spills of the caller-saved registers were
// hoisted to the hot path just the same).  It is one rolled loop BEHIND the thread's particles instead, where little is live.
// run0 + the first k + 1 weights (on the 2^-96 grid) = the exact prefix of particle k; bit k of `mask` asks for its count.
#ifndef CSSM_OFF_SELF_WAVES
#define CSSM_OFF_SELF_WAVES 5
#endif
__device__ __forceinline__ void offspring_exact_counts(cssm_u128 run0, const double (&w)[4], uint32_t mask, double totd, double u, uint64_t n_global,
                                                       uint32_t (&e)[4], bool host_weights = false) {
  const bool pow2 = (n_global & (n_global - 1)) == 0;
  const double inv_n = 1.0 / (double)n_global;
  cssm_u128 run = run0;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const double wk = (k == 0) ? w[0] : ((k == 1) ? w[1] : ((k == 2) ? w[2] : w[3]));
    run = cssm_u128_add(run, host_weights ? cssm_fix_from_double(wk) : cssm_fix_from_unit(wk));
    if ((mask >> k) & 1u) {
      const double C = cssm_u128_to_double(run) / totd;
      const uint32_t c = (uint32_t)(pow2 ? cssm_sys_count_pow2(C, u, n_global, inv_n) : cssm_sys_count(C, u, n_global));
      e[0] = (k == 0) ? c : e[0]; e[1] = (k == 1) ? c : e[1]; e[2] = (k == 2) ? c : e[2]; e[3] = (k == 3) ? c : e[3];
    }
  }
}

// Diagnostic build only (-DCSSM_OFF_STAMPS, tools/archive/offspring_stamps.py): thread 0 of every block of k_offspring_self leaves the
// constant 100 MHz clock at eight points of the kernel in the (otherwise unused) cumulative-weights buffer.
#ifdef CSSM_OFF_STAMPS
#define CSSM_STAMP(k) do { if (SELF && threadIdx.x == 0 && cum_out) reinterpret_cast<unsigned long long*>(cum_out)[(size_t)(is_pub ? gridDim.x - 1u : ublk) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSSM_STAMP(k) do { } while (0)
#endif
// The contract's correctly rounded S_tot as a double, formed only where the exact predicate is evaluated: the single-GPU launch totals
// the waves' sums its prologue left in LDS (s_r[1], not written again); the sharded launches hold the total already
template <bool SELF> struct TotExact {
  const cssm_u128* waves; double totd;
  __device__ __forceinline__ double operator()() const {
    if (!SELF) return totd;
    cssm_u128 t = waves[0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) t = cssm_u128_add(t, waves[w]);
    return cssm_u128_to_double(t);
  }
};
// End slot of a particle = cnt(C_j) of the contract.  Fast path: p = S_j/S_tot*N - u evaluated in fp64; whenever p is
// farther than eps = N*2^-44 from an integer, floor(p)+1 IS the contract's count.  Otherwise (probability 2*eps per
// particle) the exact predicate is evaluated on the exact 128-bit prefix.  Error budget of the fast path, in slots:
//   S_j as a double: the thread's exclusive prefix is converted once (two conversions: < N*2^-52), then every particle adds
//     its weight IN FLOATING POINT, sd += w1 * 2^96 (<= 4 roundings: < N*2^-51) -- the weight as a double, not its
//     truncation to the 2^-96 grid: the drift is < 4 * 2^-96 / S_tot, and S_tot >= the largest weight >= exp(-32) > 2^-46.2
//     (cssm_ref_choose admits a level at most CSSM_REF_ABOVE = 32 above the max; a level from the max itself gives 1):
//     < N*2^-47.8.  (Round 2 converted the exact 128-bit running sum per particle: 13 instructions against 1.)
//   quotient N / S_tot, the fma: < N*2^-51; the contract's own roundings move a decision by < N*2^-51.
// Total < N*2^-47.2, a factor 9 inside eps.  raw == 1 (stateless resampling of arbitrary host weights: no lower bound on
// S_tot) keeps the exact running sum.
// The thread's CSSM_ITEMS particles: run = its exclusive prefix (exact, on the 2^-96 grid), w1 its weights; e[r] = the end slot of particle r.
// totd_exact(): the contract's correctly rounded S_tot, called only where the exact predicate is evaluated.
template <bool SELF, int RS>
__device__ __forceinline__ void tile_end_slots(cssm_u128 run, const double (&w1)[CSSM_ITEMS], const int raw,
                                               const double u, const uint64_t n_global, const bool pow2, const int force_exact, const TotExact<SELF> totd_exact,
                                               const uint64_t seed, const uint32_t rec_step, const uint64_t base, const uint64_t n,
                                               double* __restrict__ cum_out, uint32_t (&e)[CSSM_ITEMS],
                                               const double scale, const double eps, const double one_minus_eps, const double one_minus_u) {
  constexpr int resampler = RS;
  auto fixw = [&](double w) { return (raw == 1) ? cssm_fix_from_double(w) : cssm_fix_from_unit(w); };
  constexpr bool OUTLINED = RS == CSSM_RESAMPLE_SYSTEMATIC && CSSM_ITEMS == 4;   // (every systematic instantiation)
  uint32_t unsafe = 0u;                                   // OUTLINED: the particles whose count the exact predicate decides
  const cssm_u128 run0 = run;
  double sd = cssm_fma((double)run.hi, 0x1.0p64, (double)run.lo);
#pragma unroll
  for (int r = 0; r < CSSM_ITEMS; ++r) {
    if (raw == 1) {
      run = cssm_u128_add(run, fixw(w1[r]));
      sd = cssm_fma((double)run.hi, 0x1.0p64, (double)run.lo);
    } else {
      sd = cssm_fma(w1[r], 0x1.0p96, sd);
    }
    // p + 1 = S_j/S_tot*N + (1 - u) > 0 (1 - u is exact: u is a multiple of 2^-53 in [0, 1)): its integer part is the count,
    // its fraction (v_fract_f64) the distance test
    const double pp1 = cssm_fma(sd, scale, one_minus_u);
    const double fr = cssm_fract_pos(pp1);
    // (OUTLINED: a forced exact evaluation does not touch this path -- the particles it names join `unsafe` behind the loop)
    const bool safe = (fr > eps) && (fr < one_minus_eps) && (OUTLINED || !force_exact) && resampler == CSSM_RESAMPLE_SYSTEMATIC;
    if (safe) {
      // (the count cannot exceed N -- the min is a guard for the ancestor writes below, not part of the arithmetic)
      const uint32_t c32 = (uint32_t)pp1;
      e[r] = (c32 > (uint32_t)n_global) ? (uint32_t)n_global : c32;
    } else if constexpr (OUTLINED) {
      e[r] = 0u;
      unsafe |= 1u << r;
    } else {
      if (raw != 1) {                                      // the exact prefix, formed only here
        run = run0;
#pragma unroll
        for (int k = 0; k < CSSM_ITEMS; ++k) if (k <= r) run = cssm_u128_add(run, fixw(w1[k]));
      }
      const double C = cssm_u128_to_double(run) / totd_exact();
      if (resampler == CSSM_RESAMPLE_SYSTEMATIC) {
        e[r] = (uint32_t)(pow2 ? cssm_sys_count_pow2(C, u, n_global, 1.0 / (double)n_global) : cssm_sys_count(C, u, n_global));
      } else if (resampler == CSSM_RESAMPLE_STRATIFIED) {   // one uniform per slot, model/Resampling.scala:82-83
        e[r] = (uint32_t)cssm_strat_count(C, seed, rec_step, n_global);
      } else {                                              // multinomial: the draws are searched in C afterwards
        const uint64_t ii = base + (uint64_t)threadIdx.x * CSSM_ITEMS + r;
        if (ii < n) cum_out[ii] = C;
        e[r] = 0;
      }
    }
  }
  if constexpr (OUTLINED) {
    // CSSM_OPT_EXACT_OFFSPRING (verification): 1 = every particle through the exact predicate, 2 = every third one -- threads
    // then hold mixed masks, as they do when the fast path hands over a single particle
    if (force_exact) {   // (uniform)
      uint32_t fm = 0xfu;
      if (force_exact == 2) {
        const uint32_t j0 = (uint32_t)base + threadIdx.x * CSSM_ITEMS;
        fm = ((j0 % 3u == 0u) ? 1u : 0u) | (((j0 + 1u) % 3u == 0u) ? 2u : 0u) | (((j0 + 2u) % 3u == 0u) ? 4u : 0u) | (((j0 + 3u) % 3u == 0u) ? 8u : 0u);
      }
      unsafe |= fm;
    }
    if (unsafe) offspring_exact_counts(run0, w1, unsafe, totd_exact(), u, n_global, e, raw == 1);
  }
}

// The end slot of ONE cumulative weight `toff` (a wave's exclusive prefix: the end slot of the particle before the wave's first): the fast
// path of tile_end_slots with its scale / eps, else the exact predicate
template <bool SELF, int RS>
__device__ __forceinline__ uint32_t end_slot_of_prefix(const cssm_u128 toff, const double scale, const double eps, const double one_minus_eps,
                                                       const double one_minus_u, const double u, const uint64_t n_global, const bool pow2,
                                                       const int force_exact, const TotExact<SELF> totd_exact, const uint64_t seed, const uint32_t rec_step) {
  constexpr int resampler = RS;
  constexpr bool OUTLINED = RS == CSSM_RESAMPLE_SYSTEMATIC && CSSM_ITEMS == 4;
  const double sdp = cssm_fma((double)toff.hi, 0x1.0p64, (double)toff.lo);
  const double ppp = cssm_fma(sdp, scale, one_minus_u);
  const double frp = cssm_fract_pos(ppp);
  if ((frp > eps) && (frp < one_minus_eps) && !force_exact && resampler == CSSM_RESAMPLE_SYSTEMATIC) {
    const uint32_t c32 = (uint32_t)ppp;
    return (c32 > (uint32_t)n_global) ? (uint32_t)n_global : c32;
  } else if constexpr (OUTLINED) {
    const double z4[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t p4[4] = {0u, 0u, 0u, 0u};
    offspring_exact_counts(toff, z4, 1u, totd_exact(), u, n_global, p4);
    return p4[0];
  } else {
    const double Cp = cssm_u128_to_double(toff) / totd_exact();
    return (resampler == CSSM_RESAMPLE_STRATIFIED)
               ? (uint32_t)cssm_strat_count(Cp, seed, rec_step, n_global)
               : (uint32_t)(pow2 ? cssm_sys_count_pow2(Cp, u, n_global, 1.0 / (double)n_global) : cssm_sys_count(Cp, u, n_global));
  }
}

// What the single-GPU launch's publisher block files once the sums are totalled and the level is known (all threads of the block call):
// the ESS of the PREVIOUS weighted observation if it was still pending (ptot2 = its sum of squared weights), this observation's max /
// level / totals / ll -- and its ESS where the squares are at hand (s2_par < 0), else the note that it is pending --, the level
// predicted for the next observation (LGCP), and the clearing of the two slot sets this observation did not use.
__device__ __forceinline__ void publish_observation(Scalars* __restrict__ sc, const StepRec* __restrict__ rec, const double gmax_dec, const double gmax,
                                                    const cssm_u128 tot, const cssm_u128 tot2, const bool p_pend, const cssm_u128 ptot2,
                                                    const int s2_par, const uint32_t nunits, const uint32_t rec_idx, const uint32_t gen,
                                                    double* __restrict__ ll_t, int32_t* __restrict__ ess_t, const uint64_t n_global, const int slot_set) {
  if (threadIdx.x == 0) {                              // publish the step's scalars once
    if (p_pend) {
      const int32_t pe = cssm_ess_of(sc->pend_S, ptot2);
      sc->ess = pe;
      if (ess_t && sc->pend_gen == gen) ess_t[sc->pend_idx] = pe;
    }
    sc->gmax = gmax_dec; sc->ref = gmax; sc->S_off = cssm_u128_zero(); sc->S_local = tot; sc->S_tot = tot;
    publish_next_level(sc, rec, gmax_dec);
    if (s2_par < 0) {
      sc->S2_local = tot2; sc->S2_tot = tot2; sc->pend = 0u;
      finish_step(sc, n_global);
      if (ll_t) { ll_t[rec_idx] = sc->ll; ess_t[rec_idx] = sc->ess; }
    } else {
      sc->pend = 1u; sc->pend_buf = (uint32_t)s2_par; sc->pend_n = nunits; sc->pend_idx = rec_idx; sc->pend_gen = gen;
      sc->pend_S = tot;
      (void)finish_ll(sc, n_global);
      if (ll_t) ll_t[rec_idx] = sc->ll;
    }
  }
  if (threadIdx.x < 2 * 2 * CSSM_GRP_MAX) {   // ... and their group sums (two sets x two limbs x 32 groups)
    const uint32_t tq2 = threadIdx.x;
    sc->grp[((size_t)((slot_set + 1 + (int)(tq2 / (2 * CSSM_GRP_MAX))) % CSSM_MAXSETS) * 2 * CSSM_GRP_MAX + tq2 % (2 * CSSM_GRP_MAX)) * CSSM_SLOT_STRIDE] = 0ull;
  }
  if (threadIdx.x < 2 * CSSM_MAXSLOTS)   // the two sets this observation did not use
    sc->maxslot[((size_t)((slot_set + 1 + (int)(threadIdx.x / CSSM_MAXSLOTS)) % CSSM_MAXSETS) * CSSM_MAXSLOTS + threadIdx.x % CSSM_MAXSLOTS) * CSSM_SLOT_STRIDE] = 0ull;
}

// What a sharded launch that has already read every rank's 5 words hands the body (k_offspring_expand_spec: one thread per rank
// loads a header, the totals go through LDS -- the body's own loops over all5 are world x 5 loads in EVERY thread, and on the
// peer-written windows each of those is a system-scope load past the caches)
struct SpecTotals { cssm_u128 S_off, tot, tot2; double gmax; };
// `Mid` (sharded launches with the group sums at hand, GRP && !SELF): what stands between the block's LOCAL work -- its first tile on the
// grid and scanned, its prefix inside the rank from the group sums: nothing of that depends on another rank -- and the rest, which
// needs every rank's totals: the wait for the peers' headers, the level check, block 0's coverage verdict.  bool operator()(SpecTotals&):
// false = the block ends here (series on hold, a peer missing); contains block barriers.  Every other instantiation passes nullptr.
struct NoMid { __device__ __forceinline__ bool operator()(SpecTotals&) const { return true; } };
template <bool FUSE, bool SELF, int RS, int RAWC = -1, int GRPL = 0, class Mid = NoMid>
__device__ __forceinline__ void offspring_body(const double* __restrict__ logw, uint64_t n,
                                                          Scalars* __restrict__ sc,
                                                          const cssm_u128* __restrict__ unitP, const cssm_u128* __restrict__ unitS2,
                                                          const StepRec* __restrict__ rec, uint64_t n_global,
                                                          uint32_t* __restrict__ endslot, uint32_t* __restrict__ anc,
                                                          uint32_t ntiles, uint32_t sup, uint32_t nunits, int raw_arg, int slot_set,
                                                          double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t rec_idx,
                                                          int force_exact, const unsigned long long* __restrict__ all5, int rank, int world,
                                                          int split, uint64_t seed, double* __restrict__ cum_out,
                                                          const double* __restrict__ logtab, int optimistic,
                                                          unsigned long long* __restrict__ flag_out,
                                                          uint32_t slot_lo, uint32_t slot_hi, uint32_t all5_stride,
                                                          cssm_u128* __restrict__ s2buf = nullptr, uint32_t s2_stride = 0, int s2_par_arg = -1,
                                                          uint32_t gen = 0, const cssm_u128* __restrict__ unit_pre = nullptr, const uint32_t blk0 = 0u,
                                                          const double* pre_in = nullptr, const SpecTotals* tt = nullptr, Mid* mid = nullptr) {
  // pre_in (or nullptr): the weights of the block's first tile, requested by the caller (the merged exchange kernel asks for them
  // BEFORE it waits for the peers' flags: the wait covers their round trip)
  // blk0: blocks [0, blk0) of the launch are somebody else's (the pack blocks of the merged exchange + offspring kernel of the
  // peer-written exchange): this body runs in blocks blk0 .. gridDim.x - 1, numbered from 0
  const uint32_t bidx = blockIdx.x - blk0, nblk = gridDim.x - blk0;
  // unit_pre (sharded, single-collective exchange; or nullptr): exclusive prefixes of the (sub-)unit sums, from k_boundary_pack
  // GRP (SELF, RAWC == 2, behind a k_propagate whose blocks accumulated them): the sums of groups of 32 units are at hand
  // (Scalars::grp) -- an instantiation of its own: it keeps ONE unit-sum entry per lane of one wave in flight instead of UPRE per
  // thread, and the registers that frees let the block's first tile be converted and scanned BEFORE the sums' barrier
  // GRPL: 0 = no group sums; 1 = layout 1 (<= 32 groups of 32 units: one wave); 2 = layout 2 (<= 64 groups of 64 units: two waves; single GPU)
  constexpr bool GRP = GRPL != 0;
  constexpr bool BIG = GRPL == 2;
  static_assert(!BIG || SELF, "layout 2: the single-GPU launch");
  constexpr bool grp_on = GRP;
  // EARLY: the block's first tile goes onto the grid and through its wave scan BEFORE the sums' barrier -- where the stored values are
  // the weights themselves (RAWC == 2); behind k_tile_sums (RAWC == 0: log-weights, rescaled by a level the block decodes first) the
  // group sums serve the prefix only
  constexpr bool EARLY = GRP && RAWC == 2;
  static_assert(!GRP || RAWC == 2 || (SELF && RAWC == 0), "group sums: a launch behind a fused-sums propagate (the single GPU's, or a shard's exchange kernel)");
  constexpr bool SHARD_GRP = GRP && !SELF;               // (the group sums are the RANK's: prefix inside the rank; totals and offset come with `mid`)
  // RAWC >= 0 (the single-GPU launches): the weight-input mode is a compile-time constant -- 2 goes with the pending ESS
  // (s2_par >= 0), 0 with the sums of squares at hand; the kernel had run out of scalar and vector registers otherwise
  const int raw = (RAWC >= 0) ? RAWC : raw_arg;
  const int s2_par = (RAWC == 0) ? -1 : s2_par_arg;
  // raw: 0 = `logw` holds log-weights, rescaled here by the step's level; 1 = weights as given (stateless Resample[A]);
  //      2 = the weights exp(min(w - c, 2^-20)) k_propagate<SUMS> stored in place of the log-weights (c = rec->ref): no exp
  //      here, and the conversion is the one k_propagate formed the unit sums with
  // all5_stride: distance in words between the 5 words of consecutive ranks (5: the all-gathered array; the segment
  // length when the words are read from the headers of the single-collective exchange, see k_boundary_pack)
  // !SELF && FUSE (sharded, stateless): only the slots [slot_lo, slot_hi) are this rank's; anc is indexed from slot_lo.
  // unitP holds `split` entries per unit (k_propagate's blocks are sub-units); all5: 5 words per rank
  // (S.lo, S.hi, S2.lo, S2.hi, order key of the rank's max); optimistic: the sums were formed relative to the
  // observation's reference level before the max was known -- if the max rules that level out, nothing is
  // resampled and the host is told to form the sums again (SELF: err bit 6, the series is on hold; else err bit 2 / *flag_out).
  // s2buf / s2_stride / s2_par / gen (SELF): two arrays of per-block partial sums of squared weights.  s2_par >= 0: this
  // launch forms the observation's sum of squares ITSELF -- block b's partial goes to s2buf[s2_par * s2_stride + b] -- and its
  // ESS stays pending (Scalars::pend) until the next weighted observation's publisher, or the host, totals the partials.
  // s2_par < 0: the sum of squares is at hand (unitS2: k_tile_sums formed it): the ESS is published with ll.
  constexpr int resampler = RS;
  const double* tab = nullptr; (void)logtab;
  __shared__ cssm_u128 s_w[CSSM_BLOCK / 64];
  __shared__ __attribute__((aligned(16))) uint32_t s_slot[FUSE ? (CSSM_BLOCK / 64) * CSSM_WAVE_CHUNK : 4];   // per wave: a 512-slot chunk of ancestor runs
  __shared__ cssm_u128 s_r[3][CSSM_BLOCK / 64];
  __shared__ cssm_u128 s_pre[2];
  __shared__ double s_scale;                               // (GRP) N / S_tot, see scan_units
  // (batch series, single GPU) an earlier observation's reference level was ruled out by its max: the series is on hold at
  // that observation until the host has redone it (run_filter_once); nothing may change meanwhile
  // (the test sits behind the prefetches below: a dependent round trip at the very top of the kernel otherwise)
  // The single-GPU launch has one block more than units: the publisher.  Where all nunits + 1 <= 1025 blocks are resident at once
  // (the default kernel: five blocks per CU) it is block 0 -- the oldest wave of its CU, served first -- and block b + 1 works on
  // unit b; the other instantiations (four blocks per CU) keep it last, where it slips into the first slot a unit block frees.
  constexpr bool PUB_FIRST = SELF && RS == CSSM_RESAMPLE_SYSTEMATIC && RAWC == 2;
  const uint32_t ublk = PUB_FIRST ? bidx - 1u : bidx;
  const bool is_pub = SELF && (PUB_FIRST ? bidx == 0u : bidx == nunits);
  // ======== stretch 1: entry -- ONE round of loads
  CSSM_STAMP(0);
  const uint32_t held = SELF ? sc->err : 0u;
  double pre_v[CSSM_ITEMS];   // the block's first tile is requested before the serial prologue
  if (pre_in != nullptr) {
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) pre_v[r] = pre_in[r];
  } else if (ublk < nunits) load_tile_raw(logw, (uint64_t)ublk * sup * CSSM_TILE, n, raw, pre_v);
  // ... and (single GPU) so are the unit sums every block totals: thread t owns the E = ceil(nsub / 256) consecutive entries
  // from t E on (up to UPRE of them in flight while the max is decoded; a loop for more)
  constexpr int UPRE = CSSM_OFF_UPRE;
  cssm_u128 upre[UPRE];
  const uint32_t nsub = SELF ? nunits * (uint32_t)split : 0u;
  const uint32_t E = (nsub + CSSM_BLOCK - 1) / CSSM_BLOCK;
  // grp_on: ONE wave (not the one that decodes the max) totals 32 group sums + the 32 unit sums of the block's own group: lane l < 32
  // holds group l (four 64-bit words of 32-bit limb sums), lane 32 + j unit j of the own group
  const uint32_t wsum = (bidx + 1u) & 3u;
  const uint32_t wsum2 = (bidx + 2u) & 3u;                 // (layout 2: the wave that scans the block's own group of 64 units)
  const uint32_t grp_unit = is_pub ? 0u : ublk;
  if (grp_on && BIG) {
#pragma unroll
    for (int k = 0; k < UPRE; ++k) upre[k] = cssm_u128_zero();
    const uint32_t l = threadIdx.x & 63u;
    if ((threadIdx.x >> 6) == wsum) {                      // lane l: group l (the two limb sums)
      const unsigned long long* g = &sc->grp[((size_t)slot_set * 2 * CSSM_GRP_MAX + l) * CSSM_SLOT_STRIDE];
      upre[0].lo = g[0]; upre[0].hi = g[(size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE];
    } else if ((threadIdx.x >> 6) == wsum2) {              // lane l: unit l of the own group
      const uint32_t q = (grp_unit / 64u) * 64u + l;
      if (q < nunits) upre[0] = unitP[q];
    }
  } else if (grp_on) {
#pragma unroll
    for (int k = 0; k < UPRE; ++k) upre[k] = cssm_u128_zero();
    if ((threadIdx.x >> 6) == wsum) {
      const uint32_t l = threadIdx.x & 63u;
      if (l < (uint32_t)CSSM_GRP_SMALL) {
        const unsigned long long* g = &sc->grp[((size_t)slot_set * 2 * CSSM_GRP_MAX + l) * CSSM_SLOT_STRIDE];
        upre[0].lo = g[0]; upre[0].hi = g[(size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE];   // the two limb sums
      } else {
        const uint32_t q = (grp_unit / CSSM_GRP_UNITS) * CSSM_GRP_UNITS + (l - (uint32_t)CSSM_GRP_SMALL);
        if (q < nunits) {
          upre[0] = unitP[(size_t)q * (uint32_t)split];
          // (a shard whose propagate ran `split` blocks per unit -- the LGCP: its sums are the blocks')
          if (!SELF) for (uint32_t sb = 1; sb < (uint32_t)split; ++sb) upre[0] = cssm_u128_add(upre[0], unitP[(size_t)q * (uint32_t)split + sb]);
        }
      }
    }
  } else if (SELF) {
    if (nsub == UPRE * CSSM_BLOCK) {   // (uniform) every cloud of 2^20 particles or more: UPRE entries per thread, none out of range
#pragma unroll
      for (int k = 0; k < UPRE; ++k) upre[k] = unitP[threadIdx.x * UPRE + (uint32_t)k];
    } else {
#pragma unroll
      for (int k = 0; k < UPRE; ++k) {
        const uint32_t q = threadIdx.x * E + (uint32_t)k;
        upre[k] = ((uint32_t)k < E && q < nsub) ? unitP[q] : cssm_u128_zero();
      }
    }
  }
  // the record's scalars are requested here, with everything else the kernel starts from (behind the max decode they were a
  // round trip of their own)
  const double rec_ref = rec->ref, u = rec->u;
  const uint32_t rec_step = rec->step;
  // ======== stretch 2: the level
  __shared__ unsigned long long s_key;
  double gmax_dec = 0.0, gmax = 0.0;
  // the level of this step, once the max is known; false: the level the sums were formed with is ruled out -- nothing may be
  // resampled (uniform: every block takes the same decision from the same words)
  auto level_known = [&]() -> bool {
    gmax = (raw == 1) ? gmax_dec : cssm_ref_choose(rec_ref, gmax_dec);
    if (raw != 1 && optimistic && !(gmax == rec_ref)) {
      if (SELF) {
        // put the series on hold AT this observation (its propagate is done: the cloud is in place, the previous ancestors are
        // untouched); every kernel enqueued behind returns at once, the host redoes this observation -- its weights again, as
        // log-weights, and its sums relative to the max -- and carries on
        if (bidx == 0 && threadIdx.x == 0) { sc->gmax = gmax_dec; atomicMin(&sc->fail_step, rec_step); atomicOr(&sc->err, 64u); }
      } else if (bidx == 0 && threadIdx.x == 0) {
        if (flag_out) *flag_out = 1ull;
        if (optimistic == 2) { atomicMin(&sc->fail_step, rec_step); atomicOr(&sc->err, 4u); }   // (merged with the expansion: no later kernel reads the flag; the series holds HERE: cssm_pf_shard_resume_level)
      }
      return false;
    }
    return true;
  };
  if (SELF) {
    // wave 0 decodes the running max (lane t reads slot t) and leaves its key in LDS; the block reads it behind the barrier of
    // the unit-sum scan below (round 2: block_decode_slots, a block barrier of its own at the head of the kernel)
    // (which wave: round-robin over the blocks -- a block's wave w runs on SIMD w, and one SIMD of every CU carrying all
    //  the decodes delayed each CU's youngest blocks at N = 2^20)
    if ((threadIdx.x >> 6) == (bidx & (CSSM_BLOCK / 64 - 1))) {
      const uint32_t l = threadIdx.x & 63u;
      unsigned long long k = (l < CSSM_MAXSLOTS) ? sc->maxslot[((size_t)slot_set * CSSM_MAXSLOTS + l) * CSSM_SLOT_STRIDE] : 0ull;
      k = wave_max_u64(k);
      if (l == 0u) s_key = k;
    }
    if (held & 64u) return;
  } else if (!SHARD_GRP) {
    if (tt != nullptr) {
      gmax_dec = tt->gmax;
    } else if (all5) {
      unsigned long long key = 0ull;
      for (int r = 0; r < world; ++r) { const unsigned long long k = ld_sys_u64(all5 + (size_t)all5_stride * r + 4); key = (k > key) ? k : key; }
      gmax_dec = cssm_order_unkey(key);
    } else {
      gmax_dec = sc->gmax;
    }
    if (!level_known()) return;
  }
  if (!SELF && !SHARD_GRP && flag_out && bidx == 0 && threadIdx.x == 0) *flag_out = 0ull;
  const bool pow2 = (n_global & (n_global - 1)) == 0;
  // (1 / N is formed where the exact predicate is evaluated: a division and two registers in every thread otherwise)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double totd = 0.0;
  cssm_u128 S_off = cssm_u128_zero();
  // sharded: the rank's offset and the global totals -- from the all-gathered sums (every block totals them itself, or the caller did: tt);
  // block 0 publishes the observation's scalars
  auto shard_totals = [&](const SpecTotals* t) {
    cssm_u128 tot = cssm_u128_zero(), tot2 = cssm_u128_zero();
    if (t != nullptr) {
      S_off = t->S_off; tot = t->tot; tot2 = t->tot2;
    } else
    for (int r = 0; r < world; ++r) {
      cssm_u128 a, b;
      const unsigned long long* w5 = all5 + (size_t)all5_stride * r;   // (possibly a peer-written window: ld_sys)
      a.lo = ld_sys_u64(w5); a.hi = ld_sys_u64(w5 + 1); b.lo = ld_sys_u64(w5 + 2); b.hi = ld_sys_u64(w5 + 3);
      if (r < rank) S_off = cssm_u128_add(S_off, a);
      tot = cssm_u128_add(tot, a); tot2 = cssm_u128_add(tot2, b);
    }
    totd = cssm_u128_to_double(tot);
    if (bidx == 0 && threadIdx.x == 0) {
      sc->gmax = gmax_dec; sc->ref = gmax;
      sc->S_off = S_off; sc->S_tot = tot; sc->S2_tot = tot2;
      finish_step(sc, n_global);
      publish_next_level(sc, rec, gmax_dec);
    }
  };
  if (!SELF && !SHARD_GRP) {
    if (all5) {
      shard_totals(tt);
    } else {
      totd = cssm_u128_to_double(sc->S_tot);
      S_off = sc->S_off;
    }
  }
  // (SELF: the single-GPU launch has exactly one block per unit -- no loop, so that what was prefetched above does not
  //  have to stay live around a back edge -- plus ONE more block, the publisher: it totals the sums like every block and
  //  then publishes the observation's scalars (ll; ess: a logarithm and two divisions in one thread) instead of working on
  //  a unit.  With one block doing both, that serial tail was on the critical path of a launch that at small N has nothing
  //  but its critical path.  Measured and dropped in round 3: letting the LAST unit's block publish when nunits + 1 blocks
  //  exceed the 4 x 256 resident slots (N = 2^20: 1024 units) -- 12.8 vs 12.0 us: blocks do not finish in lockstep, the extra
  //  block slips into the first free slot long before the grid drains.)
  // ======== stretch 3: the sums -> total + prefix; the publisher; (a shard on group sums) the peers
  uint32_t unit = ublk;
  cssm_u128 toff_self = cssm_u128_zero();
  double scale_self = 0.0;
  // (the exact S_tot as a double, where the exact predicate needs it: TotExact)
  cssm_u128 acc2 = cssm_u128_zero();                       // SELF, s2_par >= 0: the thread's sum of squared weights
  // the weight on the 2^-96 grid (raw == 1: arbitrary host doubles, range-checked; else exp of a clamped non-positive number).
  // Not kept: the rare exact path below forms it again from w1 (16 registers live across the whole tile otherwise).
  auto fixw = [&](double w) { return (raw == 1) ? cssm_fix_from_double(w) : cssm_fix_from_unit(w); };
  // (GRP instantiation) A tile up to its block barrier: the weights, their sum on the grid, the wave scan of the threads' sums (s_w[wave] = the wave's
  // total) and, behind a unit's last tile on the single GPU, the waves' sums of squared weights (s_r[2]: they ride on that barrier
  // -- the block's partial leaves behind it, its ESS is formed later; a block-wide sum of its own at the end of the kernel cost two
  // more barriers)
  auto tile_front = [&](uint32_t tile, bool first, double (&w1)[CSSM_ITEMS], bool s2_now) __attribute__((always_inline)) -> cssm_u128 {
    // (measured and dropped: requesting the NEXT tile's weights here, a register pipeline over the 16 tiles a block of the
    //  2^24 cloud walks -- 71.4 vs 71.7 us: the CU's other blocks already cover the round trip)
    if (first) weights_from_raw(pre_v, gmax, raw, w1, tab);
    else load_tile_weights(logw, (uint64_t)tile * CSSM_TILE, n, gmax, raw, w1, tab);
    cssm_u128 tsum = cssm_u128_zero();
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      tsum = cssm_u128_add(tsum, fixw(w1[r]));
      if (SELF && s2_par >= 0) acc2 = cssm_u128_add(acc2, cssm_fix_from_unit(w1[r] * w1[r]));
    }
    const cssm_u128 inc = wave_scan_u128(tsum, lane);
    if (lane == 63) s_w[wid] = inc;
    CSSM_STAMP(2);
    if (s2_now) {
      const cssm_u128 w2 = wave_scan_u128(acc2, lane);
      if (lane == 63) s_r[2][wid] = w2;
    }
    return inc;
  };
  double w1_first[CSSM_ITEMS];                             // GRP: the first tile, converted ahead of the sums' barrier
  cssm_u128 inc_first = cssm_u128_zero();
  if (SELF || SHARD_GRP) {                                 // here unitP holds the unit SUMS (k_propagate / k_tile_sums output)
    // ONE wave scan of the threads' own sums gives both the total and the prefix of the entries below this block's first
    // (qlim): that prefix = inclusive scan at thread tq - 1 + the first qlim - tq E entries of thread tq, tq = qlim / E.
    // (Round 2 took two block-wide sums with eight masked 128-bit adds per thread each: a third of the kernel's
    //  instructions at N = 2^20, where a block has one tile.)
    const uint32_t qlim = (is_pub ? 0u : unit) * (uint32_t)split;    // (the publisher needs no prefix)
    const uint32_t Ed = E ? E : 1u;                                          // (E >= 1 here: SELF; the guard is for the other instantiations)
    // (uniform; E is a power of two for every cloud of a power-of-two size: a shift instead of the division's ~25 instructions)
    const uint32_t tq = ((Ed & (Ed - 1u)) == 0u) ? (qlim >> (31 - __builtin_clz(Ed))) : qlim / Ed, rq = qlim - tq * Ed;
    auto scan_units = [&](cssm_u128& tot, cssm_u128& pre) {   // (contains one block barrier)
      if (grp_on && BIG) {                                      // (uniform) layout 2: one wave scans the 64 groups' sums, another the own group's 64 units
        if (wid == wsum) {
          cssm_u128 x, y;   // the two limb sums a0, a1 (< 2^62 each: 64 units of < 2^112) -> a0 + a1 2^56
          x.lo = upre[0].lo; x.hi = 0ull;
          y.lo = upre[0].hi << CSSM_GRP_LIMB; y.hi = upre[0].hi >> (64 - CSSM_GRP_LIMB);
          const cssm_u128 inc = wave_scan_u128(cssm_u128_add(x, y), lane);
          const uint32_t G = grp_unit / 64u;               // (uniform)
          cssm_u128 t, pg = cssm_u128_zero();
          t.lo = readlane_u64(inc.lo, 63); t.hi = readlane_u64(inc.hi, 63);
          if (G > 0u) { pg.lo = readlane_u64(inc.lo, (int)G - 1); pg.hi = readlane_u64(inc.hi, (int)G - 1); }
          if (lane == 0u) {
            s_r[1][0] = t;
#pragma unroll
            for (int w = 1; w < CSSM_BLOCK / 64; ++w) s_r[1][w] = cssm_u128_zero();
            s_pre[0] = pg;
            const double tf = cssm_fma((double)t.hi, 0x1.0p64, (double)t.lo);   // N / S_tot of the end slots' fast path, as in layout 1
            double rinv = __builtin_amdgcn_rcp(tf);
            rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
            rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
            s_scale = (double)n_global * rinv;
          }
        } else if (wid == wsum2) {
          const cssm_u128 inc = wave_scan_u128(upre[0], lane);
          const uint32_t r = grp_unit % 64u;               // (uniform)
          cssm_u128 pu = cssm_u128_zero();
          if (r > 0u) { pu.lo = readlane_u64(inc.lo, (int)r - 1); pu.hi = readlane_u64(inc.hi, (int)r - 1); }
          if (lane == 0u) s_pre[1] = pu;
        }
        __syncthreads();
        tot.lo = s_r[1][0].lo; tot.hi = s_r[1][0].hi;
        { cssm_u128 p0, p1; p0.lo = s_pre[0].lo; p0.hi = s_pre[0].hi; p1.lo = s_pre[1].lo; p1.hi = s_pre[1].hi; pre = cssm_u128_add(p0, p1); }
        return;
      }
      if (grp_on) {                                             // (uniform) one wave, one scan: groups in lanes 0..31, own group's units behind
        static_assert(CSSM_GRP_SMALL == 32 && CSSM_GRP_UNITS == 32, "one wave holds the groups and one group's units");
        if (wid == wsum) {
          cssm_u128 v = upre[0];
          if (lane < 32u) {   // the two limb sums a0, a1 (< 2^61 each) -> a0 + a1 2^56
            cssm_u128 x, y;
            x.lo = upre[0].lo; x.hi = 0ull;
            y.lo = upre[0].hi << CSSM_GRP_LIMB; y.hi = upre[0].hi >> (64 - CSSM_GRP_LIMB);
            v = cssm_u128_add(x, y);
          }
          const cssm_u128 inc = wave_scan_u128(v, lane);
          const uint32_t G = grp_unit / CSSM_GRP_UNITS, r = grp_unit % CSSM_GRP_UNITS;   // (uniform)
          cssm_u128 t, pg = cssm_u128_zero(), pu = cssm_u128_zero();
          t.lo = readlane_u64(inc.lo, 31); t.hi = readlane_u64(inc.hi, 31);
          if (G > 0u) { pg.lo = readlane_u64(inc.lo, (int)G - 1); pg.hi = readlane_u64(inc.hi, (int)G - 1); }
          if (r > 0u) {
            cssm_u128 e; e.lo = readlane_u64(inc.lo, 31 + (int)r); e.hi = readlane_u64(inc.hi, 31 + (int)r);
            pu.lo = e.lo - t.lo; pu.hi = e.hi - t.hi - (e.lo < t.lo ? 1ull : 0ull);
          }
          if (lane == 0u) {
            s_r[1][0] = t;
#pragma unroll
            for (int w = 1; w < CSSM_BLOCK / 64; ++w) s_r[1][w] = cssm_u128_zero();
            s_pre[0] = cssm_u128_add(pg, pu);
            if (SELF) {   // (a shard's group sums are its own: the total that scales its end slots comes with the peers' headers)
              // N / S_tot of the end slots' fast path (see below), once per block instead of once per thread behind the barrier
              const double tf = cssm_fma((double)t.hi, 0x1.0p64, (double)t.lo);
              double rinv = __builtin_amdgcn_rcp(tf);
              rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
              rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
              s_scale = (double)n_global * rinv;
            }
          }
        }
        __syncthreads();
        tot.lo = s_r[1][0].lo; tot.hi = s_r[1][0].hi;     // (member by member: a struct copy here stays a memcpy through a private
        pre.lo = s_pre[0].lo; pre.hi = s_pre[0].hi;       //  object, which the back end then parks in LDS: 16 bytes per thread)
        return;
      }
      cssm_u128 own = cssm_u128_zero(), part = cssm_u128_zero();
#pragma unroll
      for (int k = 0; k < UPRE; ++k) {     // (upre[k] is zero beyond E and beyond nsub)
        if ((uint32_t)k == rq && rq < (uint32_t)UPRE) part = own;   // the first rq entries of a thread (kept by thread tq only; rq is uniform)
        own = cssm_u128_add(own, upre[k]);
      }
      if (rq == (uint32_t)UPRE) part = own;
      for (uint32_t k = UPRE; k < E; ++k) {
        const uint32_t q = threadIdx.x * E + k;
        if (q < nsub) own = cssm_u128_add(own, unitP[q]);
        if (k + 1u == rq) part = own;
      }
      const cssm_u128 inc = wave_scan_u128(own, lane);
      if (lane == 63) s_r[1][wid] = inc;
      if (threadIdx.x + 1u == tq) s_pre[0] = inc;             // inclusive scan at thread tq - 1, within its wave
      if (threadIdx.x == tq) s_pre[1] = part;                 // (tq <= 255: qlim < nsub <= 256 E)
      __syncthreads();
      tot = s_r[1][0];
#pragma unroll
      for (int w = 1; w < CSSM_BLOCK / 64; ++w) tot = cssm_u128_add(tot, s_r[1][w]);
      pre = (rq > 0u) ? s_pre[1] : cssm_u128_zero();
      if (tq > 0u) {
        pre = cssm_u128_add(pre, s_pre[0]);
        const uint32_t wq = (tq - 1u) >> 6;                   // waves wholly before thread tq - 1's
#pragma unroll
        for (int w = 0; w < CSSM_BLOCK / 64 - 1; ++w) if ((uint32_t)w < wq) pre = cssm_u128_add(pre, s_r[1][w]);
      }
    };
    if (is_pub) {
      // ---- the publisher: a path of its own that ends here (sharing the scan with the unit blocks kept its partial sums
      //      alive across their whole tile loop -- 28 bytes of scratch per thread in every block)
      // It also totals what it publishes an ESS from: the unit sums of squares when they are at hand, and the squares of the
      // PREVIOUS weighted observation if its ESS is still pending.  Those partials are requested before the record that says
      // so has arrived (it is on the publisher's critical path, which at N = 2^20 is a round of its own behind 1024 resident
      // blocks): the other buffer than this launch's, one entry per unit -- what is pending whenever the previous weighted
      // observation ran this kernel; anything else is read again below.
      cssm_u128 tot2 = cssm_u128_zero(), pt2 = cssm_u128_zero();
      const uint32_t hint_buf = (s2_par >= 0) ? (uint32_t)(s2_par ^ 1) : 0u;
      {
        const cssm_u128* hb = s2buf + (size_t)hint_buf * s2_stride;
        for (uint32_t q = threadIdx.x; q < nunits; q += CSSM_BLOCK) pt2 = cssm_u128_add(pt2, hb[q]);
      }
      if (s2_par < 0) {
        cssm_u128 t2 = cssm_u128_zero();
        for (uint32_t q = threadIdx.x; q < nsub; q += CSSM_BLOCK) t2 = cssm_u128_add(t2, unitS2[q]);
        tot2 = block_sum_u128(t2, s_r[2]);
      }
      const uint32_t p_pend = sc->pend, p_buf = sc->pend_buf, p_n = sc->pend_n;
      if (p_pend && (p_buf != hint_buf || p_n != nunits)) {   // (uniform) not what was prefetched
        pt2 = cssm_u128_zero();
        const cssm_u128* pb = s2buf + (size_t)p_buf * s2_stride;
        for (uint32_t q = threadIdx.x; q < p_n; q += CSSM_BLOCK) pt2 = cssm_u128_add(pt2, pb[q]);
      }
      cssm_u128 tot, pre_unused;
      scan_units(tot, pre_unused);
      CSSM_STAMP(1);
      gmax_dec = cssm_order_unkey(s_key);
      if (!level_known()) return;
      cssm_u128 ptot2 = cssm_u128_zero();
      if (p_pend) ptot2 = block_sum_u128(pt2, s_r[2]);
      publish_observation(sc, rec, gmax_dec, gmax, tot, tot2, p_pend != 0u, ptot2, s2_par, nunits, rec_idx, gen, ll_t, ess_t, n_global, slot_set);
      CSSM_STAMP(7);
      return;
    }
    // GRP: the first tile's weights on the grid and their wave scan while the sums' wave is at work -- nothing of that depends on
    // the sums (RAWC == 2: the weights are stored as they are used); its results go through the SAME barrier (tile_front below)
    if constexpr (EARLY) {
      const uint32_t t0h = ublk * sup;
      inc_first = tile_front(t0h, true, w1_first, SELF && s2_par >= 0 && t0h + 1u == ((t0h + sup < ntiles) ? t0h + sup : ntiles));
    }
    cssm_u128 tot;
    scan_units(tot, toff_self);
    CSSM_STAMP(1);
    if constexpr (SHARD_GRP) {
      // everything local is done (first tile on the grid and scanned, the prefix inside the rank known): now the peers
      SpecTotals t2;
      if (!(*mid)(t2)) return;
      gmax_dec = t2.gmax; gmax = rec_ref;                  // (mid checked the level: the sums were formed relative to rec_ref)
      shard_totals(&t2);
    } else {
    gmax_dec = cssm_order_unkey(s_key);
    if (!level_known()) return;
    // N / S_tot for the fast path of the end slots: S_tot through two conversions and an fma, its reciprocal by v_rcp_f64 and two
    // Newton steps (what the division's own expansion starts with): relative error < 2^-50, inside the budget stated below.  The
    // correctly rounded S_tot of the contract (a normalisation with a leading-zero count: ~35 instructions, and a full division:
    // ~30, in every thread of every block) is formed only where the exact predicate is evaluated.
    if constexpr (GRP) {
      scale_self = uniform_f64(s_scale);                   // (formed by the wave that totalled the sums, ahead of the barrier)
    } else {
      const double tf = cssm_fma((double)tot.hi, 0x1.0p64, (double)tot.lo);
      double rinv = __builtin_amdgcn_rcp(tf);
      rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
      rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
      scale_self = uniform_f64((double)n_global * rinv);
    }
    }
  }
  // ======== stretches 4 + 5: the tiles of the block's unit
  if (unit < nunits) do {
    const uint32_t t0 = unit * sup;
    const uint32_t t1 = (t0 + sup < ntiles) ? t0 + sup : ntiles;
    cssm_u128 toff;                                        // cumulative weight before the current tile
    if constexpr (SHARD_GRP) {
      toff = cssm_u128_add(S_off, toff_self);              // (one unit per block: a launch with the group sums has nunits offspring blocks)
    } else
    if (!SELF && all5 && unit_pre != nullptr) {            // sharded, the prefixes of the unit sums at hand (k_boundary_pack's header block)
      // (the merged kernel's prefix block wrote them while this launch ran: system-scope loads, behind its flag)
      cssm_u128 up; const unsigned long long* upw = reinterpret_cast<const unsigned long long*>(unit_pre + (size_t)unit * split);
      up.lo = ld_sys_u64(upw); up.hi = ld_sys_u64(upw + 1);
      toff = cssm_u128_add(S_off, up);
    } else if (!SELF && all5) {                            // sharded: unitP holds the (sub-)unit SUMS; the totals came with all5
      cssm_u128 pre = cssm_u128_zero();
      const uint32_t qlim = unit * (uint32_t)split;
      for (uint32_t q = threadIdx.x; q < qlim; q += CSSM_BLOCK) pre = cssm_u128_add(pre, unitP[q]);
      pre = block_sum_u128(pre, s_r[0]);
      toff = cssm_u128_add(S_off, pre);
      __syncthreads();
    } else if (SELF) {
      toff = uniform_u128(toff_self);
    } else {
      toff = cssm_u128_add(S_off, unitP[(size_t)unit * split]);
    }
    for (uint32_t tile = t0; tile < t1; ++tile) {
      const uint64_t base = (uint64_t)tile * CSSM_TILE;
      double w1[CSSM_ITEMS];
      cssm_u128 inc;
      const bool s2_now = SELF && s2_par >= 0 && tile + 1 == t1;
      if constexpr (EARLY) {
        if (tile == t0) {                                    // (converted and scanned ahead of the sums' barrier, which covered s_w too)
#pragma unroll
          for (int r = 0; r < CSSM_ITEMS; ++r) w1[r] = w1_first[r];
          inc = inc_first;
        } else {
          inc = tile_front(tile, false, w1, s2_now);
          __syncthreads();
        }
      } else {
        // (measured and dropped: requesting the NEXT tile's weights here, a register pipeline over the 16 tiles a block of the
        //  2^24 cloud walks -- 71.4 vs 71.7 us: the CU's other blocks already cover the round trip)
        if (unit == ublk && tile == t0) weights_from_raw(pre_v, gmax, raw, w1, tab);
        else load_tile_weights(logw, base, n, gmax, raw, w1, tab);
        cssm_u128 tsum = cssm_u128_zero();
#pragma unroll
        for (int r = 0; r < CSSM_ITEMS; ++r) {
          tsum = cssm_u128_add(tsum, fixw(w1[r]));
          if (SELF && s2_par >= 0) acc2 = cssm_u128_add(acc2, cssm_fix_from_unit(w1[r] * w1[r]));
        }
        inc = wave_scan_u128(tsum, lane);
        if (lane == 63) s_w[wid] = inc;
        // (single GPU) the unit's last tile: the waves' sums of squared weights ride on this barrier -- the block's partial leaves
        // behind it, its ESS is formed later (a block-wide sum of its own at the end of the kernel cost two more barriers)
        CSSM_STAMP(2);
        if (s2_now) {
          const cssm_u128 w2 = wave_scan_u128(acc2, lane);
          if (lane == 63) s_r[2][wid] = w2;
        }
        __syncthreads();
      }
      if (s2_now && threadIdx.x == 0) {
        cssm_u128 b2 = s_r[2][0];
#pragma unroll
        for (int w = 1; w < CSSM_BLOCK / 64; ++w) b2 = cssm_u128_add(b2, s_r[2][w]);
        s2buf[(size_t)s2_par * s2_stride + ublk] = b2;
      }
      // ---- stretch 5: behind the tile's barrier
      CSSM_STAMP(3);
      cssm_u128 off = toff;
      for (int w = 0; w < wid; ++w) off = cssm_u128_add(off, s_w[w]);
      // exclusive prefix of this thread = off + the inclusive scan of the lane before (lane 0: + 0)
      cssm_u128 run = wave_excl_add_u128(inc, off);
      // end slots of the thread's particles (fast path in fp64, the exact predicate where the two could differ): tile_end_slots
      const double nd = (double)n_global;
      const double scale = SELF ? scale_self : nd / totd;
      const double eps = SELF ? uniform_f64(nd * 0x1.0p-44) : nd * 0x1.0p-44;
      const double one_minus_eps = SELF ? uniform_f64(1.0 - eps) : 1.0 - eps;
      const double one_minus_u = 1.0 - u;
      uint32_t e[CSSM_ITEMS];
      tile_end_slots<SELF, RS>(run, w1, raw, u, n_global, pow2, force_exact, TotExact<SELF>{s_r[1], totd}, seed, rec_step, base, n, cum_out, e,
                               scale, eps, one_minus_eps, one_minus_u);
      const uint64_t i0 = base + (uint64_t)threadIdx.x * CSSM_ITEMS;
      constexpr bool CLIP = !SELF;
      // the EXACT exchange of the sharded filter needs the end slots themselves (k_send_ranges, k_pack); the single-collective
      // launch passes endslot = nullptr -- nobody reads them there, and 4 bytes per particle are a third of this kernel's writes
      if ((!FUSE || (CLIP && all5 != nullptr)) && endslot != nullptr) {
        if (i0 + CSSM_ITEMS <= n) {
          *reinterpret_cast<uint4*>(endslot + i0) = make_uint4(e[0], e[1], e[2], e[3]);
        } else {
#pragma unroll
          for (int r = 0; r < CSSM_ITEMS; ++r) if (i0 + r < n) endslot[i0 + r] = e[r];
        }
      }
      CSSM_STAMP(4);
      if (FUSE && resampler != CSSM_RESAMPLE_MULTINOMIAL) {
        // end slot of the particle before this thread's first one: the lane before holds it; lane 0 of every wave evaluates
        // the count on its wave's exclusive prefix `off` -- the very sum the particle before was counted on in another wave,
        // tile or block, so the same count (round 2 passed it between the waves through LDS: one more block barrier per tile)
        uint32_t prev = dpp0<0x138 /* wave_shr:1 */, 0xf>(e[CSSM_ITEMS - 1]);
        if (lane == 0) {
          if (tile == 0 && wid == 0 && (SELF || all5 == nullptr || rank == 0)) prev = 0u;   // the globally first particle
          else prev = end_slot_of_prefix<SELF, RS>(off, scale, eps, one_minus_eps, one_minus_u, u, n_global, pow2, force_exact, TotExact<SELF>{s_r[1], totd}, seed, rec_step);
        }
        // the slots this WAVE's 256 particles own: [start of its first particle's run, end of its last particle's run), clipped
        // to this launch's slots; their ancestors are assembled in the wave's LDS region and written as whole lines, with no
        // block barrier (fill_runs_wave)
        uint32_t wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
        uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)e[CSSM_ITEMS - 1], 63);
        if (CLIP) { wb = (wb < slot_lo) ? slot_lo : wb; we = (we > slot_hi) ? slot_hi : we; }
        we = (we > (uint32_t)n_global) ? (uint32_t)n_global : we;
        wb = (wb > we) ? we : wb;
        fill_runs_wave<CSSM_OFF_SC1 != 0, CLIP>(prev, e, (uint32_t)i0, wb, we, anc, CLIP ? slot_lo : 0u, (uint32_t)(n - 1),
                                                s_slot + wid * CSSM_WAVE_CHUNK);
      }
      CSSM_STAMP(5);
      // advance the running prefix by this tile's total (not behind a unit's last tile on the single GPU: nothing follows)
      if (!SELF || tile + 1 < t1) {
        cssm_u128 ttot = s_w[0];
#pragma unroll
        for (int w = 1; w < CSSM_BLOCK / 64; ++w) ttot = cssm_u128_add(ttot, s_w[w]);
        toff = SELF ? uniform_u128(cssm_u128_add(toff, ttot)) : cssm_u128_add(toff, ttot);
        __syncthreads();                                       // (s_w is rewritten by the next tile / the next unit)
      }
    }
  } while (!SELF && !SHARD_GRP && (unit += nblk) < nunits);
}

#define CSSM_OFFSPRING_PARAMS                                                                                              \
  const double* __restrict__ logw, uint64_t n, Scalars* __restrict__ sc, const cssm_u128* __restrict__ unitP,              \
  const cssm_u128* __restrict__ unitS2, const StepRec* __restrict__ rec, uint64_t n_global, uint32_t* __restrict__ endslot, \
  uint32_t* __restrict__ anc, uint32_t ntiles, uint32_t sup, uint32_t nunits, int raw, int slot_set,                        \
  double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t rec_idx, int force_exact,                                \
  const unsigned long long* __restrict__ all5, int rank, int world, int split, uint64_t seed, double* __restrict__ cum_out, \
  const double* __restrict__ logtab, int optimistic, unsigned long long* __restrict__ flag_out, uint32_t slot_lo, uint32_t slot_hi
#define CSSM_OFFSPRING_FWD                                                                                                   \
  logw, n, sc, unitP, unitS2, rec, n_global, endslot, anc, ntiles, sup, nunits, raw, slot_set, ll_t, ess_t, rec_idx,       \
  force_exact, all5, rank, world, split, seed, cum_out, logtab, optimistic, flag_out, slot_lo, slot_hi

template <bool FUSE, bool SELF, int RS>
__global__ __launch_bounds__(CSSM_BLOCK, CSSM_OFF_WAVES) void k_offspring(CSSM_OFFSPRING_PARAMS, uint32_t all5_stride = 5) {
  offspring_body<FUSE, SELF, RS>(CSSM_OFFSPRING_FWD, all5_stride);
}

// The single-GPU filter's launch: only the arguments that path uses (the generic kernel above carries ~30, most of them the
// sharded filter's; their scalar registers spilled into vector registers and those into scratch -- 28 bytes per thread,
// i.e. 7 MB of scratch write-back per launch at N = 2^20, which the PMC counters showed as "wasted" write traffic).
template <int RS, int RAWC, int GRP = 0>
__global__ __attribute__((amdgpu_flat_work_group_size(CSSM_BLOCK, CSSM_BLOCK),
                          amdgpu_waves_per_eu((RS == CSSM_RESAMPLE_SYSTEMATIC && RAWC == 2) ? CSSM_OFF_SELF_WAVES : CSSM_OFF_WAVES, 8))) void k_offspring_self(
    const double* __restrict__ logw, uint64_t n, Scalars* __restrict__ sc, const cssm_u128* __restrict__ unitP,
    const cssm_u128* __restrict__ unitS2, const StepRec* __restrict__ rec, uint32_t* __restrict__ anc, uint32_t ntiles, uint32_t sup,
    uint32_t nunits, int slot_set, double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t rec_idx, int force_exact, int split,
    uint64_t seed, double* __restrict__ cum_out, cssm_u128* __restrict__ s2buf, uint32_t s2_stride, int s2_par, uint32_t gen) {
  // RAWC = 2: behind k_propagate<SUMS> (weights in place of log-weights, sums relative to the reference level: `optimistic`,
  // the ESS stays pending); 0: behind k_tile_sums (log-weights, both sums at hand)
  // GRP: the propagate behind this launch accumulated the sums of groups of units (Scalars::grp)
  offspring_body<true, true, RS, RAWC, GRP>(logw, n, sc, unitP, unitS2, rec, n, nullptr, anc, ntiles, sup, nunits, RAWC, slot_set, ll_t, ess_t, rec_idx,
                                            force_exact, nullptr, 0, 1, split, seed, cum_out, nullptr, RAWC == 2 ? 1 : 0, nullptr, 0u, (uint32_t)n, 5u,
                                            s2buf, s2_stride, s2_par, gen);
}


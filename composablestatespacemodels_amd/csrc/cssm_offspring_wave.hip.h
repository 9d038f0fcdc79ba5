// cssm_offspring_wave.hip.h -- k_offspring_wave: systematic resampling of a single-GPU cloud behind a fused-sums propagate whose WAVES
// own contiguous quarter-units and left their exact sums behind (round 6; Resampling.scala:36-72 as cssm_offspring.hip.h).
//
// What differs from k_offspring_self<SYSTEMATIC, 2, GRP>, whose result it reproduces bit for bit:
//   * no weight is converted to the 2^-96 grid for the prefix on the hot path.  k_propagate's tile-after-tile instantiation gives wave w
//     of a unit's block the particles [w q, (w + 1) q) of the unit (q = a quarter unit) and stores the wave's exact fixed-point sum
//     (unitW, 16 bytes per quarter unit: its wave reduction existed already).  A wave of this kernel owns the same quarter: its exclusive
//     prefix is exact integers -- the unit's prefix from the group sums + the quarters before it -- and everything behind that prefix is
//     fp64: thread sums, a 64-bit DPP scan (18 instructions against the 128-bit scan's 24 + 40 of conversions), a running chunk prefix.
//   * the end slots' fast path is the one of tile_end_slots: p + 1 = S_j / S_tot * N + (1 - u) in fp64; its integer part IS the contract's
//     count whenever its fraction is farther than eps from 0 and 1.  Error budget in slots (N = particles, S in weight units):
//       exact prefix -> double: two conversions + fma           N 2^-52
//       running chunk prefix: <= 31 adds (a unit has <= 32 tiles)   N 31 2^-53
//       thread sum (3), scan (6), + prefix (1), per particle (<= 4)  N 14 2^-53
//       N / S_tot (rcp + two Newton steps) 2^-50, the fma 2^-53, the contract's own roundings 2^-51
//     together < N 2^-47.1, a factor 8 inside eps = N 2^-44.  The fp64 sums add the weights THEMSELVES, the contract their truncation to
//     the grid: < 2^-96 per particle, < q 2^-96 over a wave's quarter, i.e. N q 2^-96 / S_tot slots -- nothing unless S_tot is tiny (the
//     level may sit 32 above the max: S_tot >= 2^-46.2), so eps carries the term: eps = N (2^-44 + 2 q 2^-96 / S_tot).
//   * where a fraction of a chunk falls inside the band (2 eps of the particles: 2^-19 at N = 2^24, one chunk of 256 in 2^11) the WAVE notes
//     the chunk and redoes it BEHIND its loop, where nothing of the hot path is live: the chunk's exact prefix (the quarter's exact prefix
//     + the chunk sums before it, re-read and converted), a 128-bit scan of the chunk and offspring_exact_counts for every particle of it
//     -- the contract's predicate on the contract's sums, as everywhere else.  (Resolved in place, inside the loop, that path's registers
//     took the kernel from 63 vector registers to 110; as a real call its caller-saved spills were hoisted onto the hot path.)
//   * one block barrier per block (behind the group sums' scan) instead of one + one or two per tile; the waves' sums of squared weights
//     meet through an LDS ticket, not a barrier.
#pragma once

#include "cssm_offspring.hip.h"

// A DPP move of all rows whose lanes without a source read 0 (bound_ctrl): no `old` operand to prepare -- dpp0's zero-initialised
// destination costs two v_mov per 64-bit move, which in the six steps of a scan is a third of its instructions
template <int CTRL>
__device__ __forceinline__ uint64_t dppz_u64(uint64_t v) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, true) |
         ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true) << 32);
}
// inclusive scan of doubles across the 64 lanes (a lane without a DPP source adds +0.0)
__device__ __forceinline__ double wave_scan_f64(double v) {
#define CSSM_F64_STEPZ(CTRL) { const double o = cssm_u2d(dppz_u64<CTRL>(cssm_d2u(v))); v = v + o; }
#define CSSM_F64_STEP(CTRL, RM) { const double o = cssm_u2d(dpp0_u64<CTRL, RM>(cssm_d2u(v))); v = v + o; }
  CSSM_F64_STEPZ(CSSM_DPP_ROW_SHR(1)) CSSM_F64_STEPZ(CSSM_DPP_ROW_SHR(2)) CSSM_F64_STEPZ(CSSM_DPP_ROW_SHR(4))
  CSSM_F64_STEPZ(CSSM_DPP_ROW_SHR(8)) CSSM_F64_STEP(CSSM_DPP_BCAST15, 0xa) CSSM_F64_STEP(CSSM_DPP_BCAST31, 0xc)
#undef CSSM_F64_STEP
#undef CSSM_F64_STEPZ
  return v;
}
// a + b on values the compiler may keep in scalar registers (cssm_u128_add's carry chain is vector assembly)
__device__ __forceinline__ cssm_u128 u128_add_any(cssm_u128 a, cssm_u128 b) {
  cssm_u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}
__device__ __forceinline__ double u128_to_f64_fast(cssm_u128 a) { return cssm_fma((double)a.hi, 0x1.0p64, (double)a.lo); }

#ifndef CSSM_OFFW_WAVES
#define CSSM_OFFW_WAVES 6
#endif

// GRPL: the layout of the group sums the propagate behind this launch added to (1: <= 32 groups of 32 units, one wave scans groups and
// the own group's units; 2: <= 64 groups of 64 units, two waves).  unitW: 4 entries per unit, the exact sums of its quarter units.
template <int GRPL>
__global__ __attribute__((amdgpu_flat_work_group_size(CSSM_BLOCK, CSSM_BLOCK), amdgpu_waves_per_eu(CSSM_OFFW_WAVES, 8))) void k_offspring_wave(
    const double* __restrict__ logw, uint64_t n, Scalars* __restrict__ sc, const cssm_u128* __restrict__ unitP, const cssm_u128* __restrict__ unitW,
    const StepRec* __restrict__ rec, uint32_t* __restrict__ anc, uint32_t sup, uint32_t nunits, int slot_set,
    double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t rec_idx, int force_exact,
    cssm_u128* __restrict__ s2buf, uint32_t s2_stride, int s2_par, uint32_t gen) {
  static_assert(GRPL == 1 || GRPL == 2, "group-sum layout 1 or 2");
  static_assert(CSSM_BLOCK == 256 && CSSM_ITEMS == 4, "four waves of 256 particles per chunk");
  constexpr bool BIG = GRPL == 2;
  __shared__ __attribute__((aligned(16))) uint32_t s_slot[(CSSM_BLOCK / 64) * CSSM_WAVE_CHUNK];
  __shared__ cssm_u128 s_tot, s_pre[2], s_r2[CSSM_BLOCK / 64], s_red[CSSM_BLOCK / 64];
  __shared__ double s_scale;
  __shared__ unsigned long long s_key;
  __shared__ uint32_t s_cnt;
  const uint32_t bidx = blockIdx.x;
  const bool is_pub = bidx == 0u;                              // block 0 publishes the observation; block b + 1 resamples unit b
  const uint32_t ublk = bidx - 1u;
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  // ======== entry: one round of loads
  const uint32_t held = sc->err;
  // (indices are 32-bit: the library admits n <= 2^32 - 2^16 particles per handle, and a cloud's last unit ends below n + 2^15)
  const uint32_t n32 = (uint32_t)n;
  const uint32_t q = sup * (uint32_t)(CSSM_TILE / 4);          // particles of a quarter unit: what one wave owns
  const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ublk * sup * (uint32_t)CSSM_TILE + wid * q));   // the wave's first particle
  double w1[CSSM_ITEMS];
  auto load_chunk = [&](uint32_t base, double (&v)[CSSM_ITEMS]) {          // the wave's 256 particles from `base` on, four per lane
    const uint32_t i0 = base + lane * CSSM_ITEMS;
    if (i0 + CSSM_ITEMS <= n32) {
      const double2 a = *reinterpret_cast<const double2*>(logw + i0);
      const double2 b = *reinterpret_cast<const double2*>(logw + i0 + 2);
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) v[r] = (i0 + r < n32) ? logw[i0 + r] : 0.0;
    }
  };
#ifdef CSSM_OFFW_WEIGHTS_FIRST
  if (!is_pub) load_chunk(w_lo, w1);
#endif
  // the sums one wave scans for the block (which wave: round-robin over the blocks, as in k_offspring_self)
  const uint32_t wsum = (bidx + 1u) & 3u, wsum2 = (bidx + 2u) & 3u, wkey = bidx & 3u;
  const uint32_t grp_unit = is_pub ? 0u : ublk;
  cssm_u128 gsum = cssm_u128_zero();
  if (BIG) {
    if (wid == wsum) {                                         // lane l: group l (the two limb sums)
      const unsigned long long* g = &sc->grp[((size_t)slot_set * 2 * CSSM_GRP_MAX + lane) * CSSM_SLOT_STRIDE];
      gsum.lo = g[0]; gsum.hi = g[(size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE];
    } else if (wid == wsum2) {                                 // lane l: unit l of the own group
      const uint32_t qq = (grp_unit / 64u) * 64u + lane;
      if (qq < nunits) gsum = unitP[qq];
    }
  } else if (wid == wsum) {
    if (lane < (uint32_t)CSSM_GRP_SMALL) {
      const unsigned long long* g = &sc->grp[((size_t)slot_set * 2 * CSSM_GRP_MAX + lane) * CSSM_SLOT_STRIDE];
      gsum.lo = g[0]; gsum.hi = g[(size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE];
    } else {
      const uint32_t qq = (grp_unit / CSSM_GRP_UNITS) * CSSM_GRP_UNITS + (lane - (uint32_t)CSSM_GRP_SMALL);
      if (qq < nunits) gsum = unitP[qq];
    }
  }
  // the quarter units before this wave's (lanes 0 .. 3 hold the unit's four; one 64-byte line)
  cssm_u128 wq = cssm_u128_zero();
  if (!is_pub) wq = unitW[(size_t)ublk * 4u + (lane & 3u)];
  // ... and only then the wave's first 256 weights: a wave's loads come back in the order they were issued, and the block's critical
  // path is its sums' wave (loads -> 128-bit scan -> LDS -> the barrier every wave waits at)
#ifndef CSSM_OFFW_WEIGHTS_FIRST
  if (!is_pub) load_chunk(w_lo, w1);
#endif
  const double rec_ref = rec->ref, u = rec->u;
  const uint32_t rec_step = rec->step;
  if (wid == wkey) {                                           // the running max: lane t reads slot t
    unsigned long long k = (lane < CSSM_MAXSLOTS) ? sc->maxslot[((size_t)slot_set * CSSM_MAXSLOTS + lane) * CSSM_SLOT_STRIDE] : 0ull;
    k = wave_max_u64(k);
    if (lane == 0u) s_key = k;
  }
  if (threadIdx.x == 0) s_cnt = 0u;
  if (held & 64u) return;                                      // (an earlier observation of the series is on hold: nothing may change)
  // ======== ahead of the barrier: everything of the first chunk that needs no sum -- thread sums, their scan, the squares
  cssm_u128 acc2 = cssm_u128_zero();
  auto squares = [&](const double (&v)[CSSM_ITEMS]) {
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) acc2 = cssm_u128_add(acc2, cssm_fix_from_unit(v[r] * v[r]));
  };
  double inc = 0.0;
  if (!is_pub) {
    inc = wave_scan_f64((w1[0] + w1[1]) + (w1[2] + w1[3]));
    squares(w1);
  }
  // ======== the sums: total, the unit's prefix, N / S_tot (one wave; layout 2: two)
  if (BIG) {
    if (wid == wsum) {
      cssm_u128 x, y;   // the two limb sums a0, a1 -> a0 + a1 2^56
      x.lo = gsum.lo; x.hi = 0ull;
      y.lo = gsum.hi << CSSM_GRP_LIMB; y.hi = gsum.hi >> (64 - CSSM_GRP_LIMB);
      const cssm_u128 sc_inc = wave_scan_u128(cssm_u128_add(x, y), (int)lane);
      const uint32_t G = grp_unit / 64u;
      cssm_u128 t, pg = cssm_u128_zero();
      t.lo = readlane_u64(sc_inc.lo, 63); t.hi = readlane_u64(sc_inc.hi, 63);
      if (G > 0u) { pg.lo = readlane_u64(sc_inc.lo, (int)G - 1); pg.hi = readlane_u64(sc_inc.hi, (int)G - 1); }
      if (lane == 0u) {
        s_tot = t; s_pre[0] = pg;
        const double tf = u128_to_f64_fast(t);
        double rinv = __builtin_amdgcn_rcp(tf);
        rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
        rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
        s_scale = (double)n * rinv;
      }
    } else if (wid == wsum2) {
      const cssm_u128 sc_inc = wave_scan_u128(gsum, (int)lane);
      const uint32_t r = grp_unit % 64u;
      cssm_u128 pu = cssm_u128_zero();
      if (r > 0u) { pu.lo = readlane_u64(sc_inc.lo, (int)r - 1); pu.hi = readlane_u64(sc_inc.hi, (int)r - 1); }
      if (lane == 0u) s_pre[1] = pu;
    }
  } else if (wid == wsum) {
    static_assert(CSSM_GRP_SMALL == 32 && CSSM_GRP_UNITS == 32, "one wave holds the groups and one group's units");
    cssm_u128 v = gsum;
    if (lane < 32u) {
      cssm_u128 x, y;
      x.lo = gsum.lo; x.hi = 0ull;
      y.lo = gsum.hi << CSSM_GRP_LIMB; y.hi = gsum.hi >> (64 - CSSM_GRP_LIMB);
      v = cssm_u128_add(x, y);
    }
    const cssm_u128 sc_inc = wave_scan_u128(v, (int)lane);
    const uint32_t G = grp_unit / CSSM_GRP_UNITS, r = grp_unit % CSSM_GRP_UNITS;
    cssm_u128 t, pg = cssm_u128_zero(), pu = cssm_u128_zero();
    t.lo = readlane_u64(sc_inc.lo, 31); t.hi = readlane_u64(sc_inc.hi, 31);
    if (G > 0u) { pg.lo = readlane_u64(sc_inc.lo, (int)G - 1); pg.hi = readlane_u64(sc_inc.hi, (int)G - 1); }
    if (r > 0u) {
      cssm_u128 e; e.lo = readlane_u64(sc_inc.lo, 31 + (int)r); e.hi = readlane_u64(sc_inc.hi, 31 + (int)r);
      pu.lo = e.lo - t.lo; pu.hi = e.hi - t.hi - (e.lo < t.lo ? 1ull : 0ull);
    }
    if (lane == 0u) {
      s_tot = t; s_pre[0] = pg; s_pre[1] = pu;
      const double tf = u128_to_f64_fast(t);
      double rinv = __builtin_amdgcn_rcp(tf);
      rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
      rinv = cssm_fma(cssm_fma(-tf, rinv, 1.0), rinv, rinv);
      s_scale = (double)n * rinv;
    }
  }
  // ---- the publisher totals what it files an ESS from while the sums' wave works: the squares of the PREVIOUS weighted observation
  cssm_u128 pt2 = cssm_u128_zero();
  uint32_t p_pend = 0u, p_buf = 0u, p_n = 0u;
  if (is_pub) {
    const uint32_t hint_buf = (uint32_t)(s2_par ^ 1);
    const cssm_u128* hb = s2buf + (size_t)hint_buf * s2_stride;
    for (uint32_t qq = threadIdx.x; qq < nunits; qq += CSSM_BLOCK) pt2 = cssm_u128_add(pt2, hb[qq]);
    p_pend = sc->pend; p_buf = sc->pend_buf; p_n = sc->pend_n;
    if (p_pend && (p_buf != hint_buf || p_n != nunits)) {      // (uniform) not what was prefetched
      pt2 = cssm_u128_zero();
      const cssm_u128* pb = s2buf + (size_t)p_buf * s2_stride;
      for (uint32_t qq = threadIdx.x; qq < p_n; qq += CSSM_BLOCK) pt2 = cssm_u128_add(pt2, pb[qq]);
    }
  }
  __syncthreads();
  // ======== the level (every block takes the same decision from the same words)
  const double gmax_dec = cssm_order_unkey(s_key);
  const double gmax = cssm_ref_choose(rec_ref, gmax_dec);
  if (!(gmax == rec_ref)) {
    // the max rules the level out the sums were formed with: the series goes on hold AT this observation, the host redoes it
    if (is_pub && threadIdx.x == 0) { sc->gmax = gmax_dec; atomicMin(&sc->fail_step, rec_step); atomicOr(&sc->err, 64u); }
    return;
  }
  cssm_u128 tot;
  tot.lo = s_tot.lo; tot.hi = s_tot.hi;
  if (is_pub) {
    cssm_u128 ptot2 = cssm_u128_zero();
    if (p_pend) ptot2 = block_sum_u128(pt2, s_red);
    publish_observation(sc, rec, gmax_dec, gmax, tot, cssm_u128_zero(), p_pend != 0u, ptot2, s2_par, nunits, rec_idx, gen, ll_t, ess_t, n, slot_set);
    return;
  }
  // ======== the wave's quarter unit, chunk after chunk; nothing below waits for another wave
  const double nd = (double)n32;
  const double scale = uniform_f64(uniform_f64(s_scale) * 0x1.0p96);        // N / S_tot with S_tot in weight units
  cssm_u128 toff;                                              // exact: everything before this wave's first particle
  {
    cssm_u128 p0, p1; p0.lo = s_pre[0].lo; p0.hi = s_pre[0].hi; p1.lo = s_pre[1].lo; p1.hi = s_pre[1].hi;
    toff = u128_add_any(p0, p1);
#pragma unroll
    for (int w = 0; w < CSSM_BLOCK / 64 - 1; ++w) {
      cssm_u128 e; e.lo = readlane_u64(wq.lo, w); e.hi = readlane_u64(wq.hi, w);
      if ((uint32_t)w < wid) toff = u128_add_any(toff, e);
    }
    toff = uniform_u128(toff);
  }
  // eps = N (2^-44 + 2 q 2^-96 / S_tot): 1 / S_tot = scale / N
  const double eps = uniform_f64(cssm_fma((double)(2u * q) * 0x1.0p-96, scale, nd * 0x1.0p-44));
  const double one_minus_eps = uniform_f64(1.0 - eps);
  const double one_minus_u = uniform_f64(1.0 - u);
  double pre_w = uniform_f64(u128_to_f64_fast(toff) * 0x1.0p-96);   // the running prefix of the wave's chunks, weight units
  const uint32_t w_hi = (w_lo + q < n32 && w_lo + q > w_lo) ? w_lo + q : n32;
  uint32_t cold = 0u;                                         // bit c: chunk c of the quarter waits for the exact path (a unit has <= 32 tiles)
  for (uint32_t base = w_lo; base < w_hi; base += CSSM_TILE / 4) {
    // the NEXT chunk's weights travel while this one is resolved (no barrier stands between a wave's chunks any more: what hid the round
    // trip in k_offspring_self -- the CU's other blocks -- is now worth having in flight)
    const bool more = base + (uint32_t)(CSSM_TILE / 4) < w_hi;
    double wn[CSSM_ITEMS];
    if (more) load_chunk(base + (uint32_t)(CSSM_TILE / 4), wn);
    // exclusive prefix of the thread's first particle: the lane before's inclusive sum (lane 0: nothing) on the chunk's prefix
    double sd = pre_w + cssm_u2d(dppz_u64<0x138 /* wave_shr:1 */>(cssm_d2u(inc)));
    uint32_t e[CSSM_ITEMS];
    uint32_t unsafe = 0u;
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      sd = sd + w1[r];
      const double pp1 = cssm_fma(sd, scale, one_minus_u);
      const double fr = cssm_fract_pos(pp1);
      const uint32_t c32 = (uint32_t)pp1;
      e[r] = (c32 > n32) ? n32 : c32;
      if (!((fr > eps) && (fr < one_minus_eps))) unsafe |= 1u << r;
    }
    // the end slot of the particle before the wave's first: counted on the chunk's prefix -- the very sum that particle was counted on
    // in another wave or block, so the same count
    uint32_t prev = dpp0<0x138 /* wave_shr:1 */, 0xf>(e[CSSM_ITEMS - 1]);
    bool prev_unsafe = false;
    if (lane == 0u) {
      if (base == 0u) prev = 0u;                             // the globally first particle
      else {
        const double ppp = cssm_fma(pre_w, scale, one_minus_u);
        const double frp = cssm_fract_pos(ppp);
        const uint32_t c32 = (uint32_t)ppp;
        prev = (c32 > n32) ? n32 : c32;
        prev_unsafe = !((frp > eps) && (frp < one_minus_eps));
      }
    }
    // A chunk with a fraction inside the band (or under CSSM_OPT_EXACT_OFFSPRING, verification: every chunk) is NOT resolved here: it is
    // noted and redone behind the loop through the contract's predicate on exact sums (see below) -- inlined at this point that path's
    // registers (128-bit sums, a division, the counting loops) took the kernel from 63 vector registers to 110.
    const bool deferred = force_exact != 0 || __any(unsafe != 0u || prev_unsafe);
    if (deferred) cold |= 1u << ((base - w_lo) >> 8);
    else {
    // the slots this wave's 256 particles own, assembled in the wave's LDS region and written as whole lines
    uint32_t wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
    uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)e[CSSM_ITEMS - 1], 63);
    we = (we > n32) ? n32 : we;
    wb = (wb > we) ? we : wb;
    fill_runs_wave<CSSM_OFF_SC1 != 0, false>(prev, e, base + lane * CSSM_ITEMS, wb, we, anc, 0u, n32 - 1u, s_slot + wid * CSSM_WAVE_CHUNK);
    }
    // advance the running prefix by the chunk's total (the scan's last lane); the next chunk: sums, scan, squares
    pre_w = uniform_f64(pre_w + cssm_u2d(readlane_u64(cssm_d2u(inc), 63)));
    if (more) {
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) w1[r] = wn[r];
      inc = wave_scan_f64((w1[0] + w1[1]) + (w1[2] + w1[3]));
      squares(w1);
    }
  }
  // ======== the block's partial of the observation's sum of squared weights: the waves meet through a ticket, the last one files it
  {
    const cssm_u128 w2 = wave_sum_u128(acc2);
    if (lane == 0u) {
      s_r2[wid] = w2;
      const uint32_t tk = __hip_atomic_fetch_add(&s_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (tk == (uint32_t)(CSSM_BLOCK / 64 - 1)) {
        cssm_u128 b2 = s_r2[0];
#pragma unroll
        for (int w = 1; w < CSSM_BLOCK / 64; ++w) b2 = cssm_u128_add(b2, s_r2[w]);
        s2buf[(size_t)s2_par * s2_stride + ublk] = b2;
      }
    }
  }
  // ======== the chunks the loop left: the contract's predicate on the contract's sums for every particle of the chunk (wave-uniform;
  //          2 eps x 256 of the chunks: 2^-11 at N = 2^24) -- nothing of the hot path is live any more
  while (cold != 0u) {
    const uint32_t c = (uint32_t)__builtin_ctz(cold);
    cold &= cold - 1u;
    const uint32_t base = w_lo + c * (uint32_t)(CSSM_TILE / 4);
    cssm_u128 X = toff;                                        // exact: everything before the chunk's first particle
    for (uint32_t b2 = w_lo; b2 < base; b2 += CSSM_TILE / 4) {
      double v[CSSM_ITEMS];
      load_chunk(b2, v);
      cssm_u128 ts = cssm_u128_zero();
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) ts = cssm_u128_add(ts, cssm_fix_from_unit(v[r]));
      X = u128_add_any(X, wave_sum_u128(ts));
    }
    load_chunk(base, w1);
    cssm_u128 ts = cssm_u128_zero();
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) ts = cssm_u128_add(ts, cssm_fix_from_unit(w1[r]));
    const cssm_u128 run0 = wave_excl_add_u128(wave_scan_u128(ts, (int)lane), X);
    cssm_u128 tote; tote.lo = s_tot.lo; tote.hi = s_tot.hi;
    const double totd = cssm_u128_to_double(tote);
    uint32_t e[CSSM_ITEMS] = {0u, 0u, 0u, 0u};
    offspring_exact_counts(run0, w1, 0xfu, totd, u, (uint64_t)n32, e);
    uint32_t prev = dpp0<0x138 /* wave_shr:1 */, 0xf>(e[CSSM_ITEMS - 1]);
    if (lane == 0u) {
      prev = 0u;                                               // (the globally first particle)
      if (base != 0u) {
        const double z4[4] = {0.0, 0.0, 0.0, 0.0};
        uint32_t p4[4] = {0u, 0u, 0u, 0u};
        offspring_exact_counts(X, z4, 1u, totd, u, (uint64_t)n32, p4);
        prev = p4[0];
      }
    }
    uint32_t wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
    uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)e[CSSM_ITEMS - 1], 63);
    we = (we > n32) ? n32 : we;
    wb = (wb > we) ? we : wb;
    fill_runs_wave<CSSM_OFF_SC1 != 0, false>(prev, e, base + lane * CSSM_ITEMS, wb, we, anc, 0u, n32 - 1u, s_slot + wid * CSSM_WAVE_CHUNK);
  }
}

// cssm_kernels.hip.h -- the kernels of libcssm_pf other than k_propagate (cssm_propagate.hip.h) that the single-GPU drivers use, and
// the bodies they share with the sharded stages (cssm_shard_kernels.hip.h).  Non-template kernels have internal linkage: the
// header is included by cssm_pf.hip and cssm_shard.hip.
#pragma once

#include "cssm_device.hip.h"

// ------------------------------------------------------------------------------------ init

// initialiseState, model/ParticleFilter.scala:105-108: x0 = sqrt(c0) z + m0
template <int D>
__global__ __launch_bounds__(CSSM_BLOCK) void k_init(double* __restrict__ dst, size_t stride, uint64_t n,
                                                     uint64_t gid0, uint64_t seed, const double* __restrict__ m0,
                                                     const double* __restrict__ sd0, const double* __restrict__ logtab) {
  const double* tab = stage_log_table(logtab);
  for (uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CSSM_BLOCK) {
    double z[D];
    draw_normals<D>(seed, gid0 + i, 0u, CSSM_STREAM_INIT, tab, z);
#pragma unroll
    for (int k = 0; k < D; ++k) dst[(size_t)k * stride + i] = sd0[k] * z[k] + m0[k];
  }
}

// FilterInit.initialiseState, model/ParticleFilter.scala:257-260
static __global__ void k_init_from(double* __restrict__ dst, size_t stride, uint64_t n, int d, const double* __restrict__ s) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    for (int k = 0; k < d; ++k) dst[(size_t)k * stride + i] = s[k];
}

// ------------------------------------------------------------------------------------ tile sums

// (load_tile_raw / weights_from_raw / load_tile_weights: cssm_device.hip.h)

// w1 = exp(w - max) (model/ParticleFilter.scala:125); S = sum w1, S2 = sum w1^2, fixed point, one pair
// per UNIT of `sup` consecutive tiles (sup is chosen on the host so that there are ~1K units: the
// scan over units then fits one pass of one block; integer sums make any grouping give the same bits).
static __global__ __launch_bounds__(CSSM_BLOCK) void k_tile_sums(const double* __restrict__ logw, uint64_t n,
                                                          Scalars* __restrict__ sc,
                                                          cssm_u128* __restrict__ unitS, cssm_u128* __restrict__ unitS2,
                                                          uint32_t ntiles, uint32_t sup, uint32_t nunits, int raw, int slot_set,
                                                          const double* __restrict__ gmax_in, const double* __restrict__ logtab,
                                                          const StepRec* __restrict__ rec, uint32_t hold_mask,
                                                          const unsigned long long* __restrict__ all5 = nullptr, int world = 0, int grp_arg = 0) {
  // grp_arg (single GPU; bits as k_propagate's set argument: 8 = on, 9-10 the set, 11-12 log2(units per group / 32)): the units' sums
  // are added to their groups' as the fused kernel does -- k_offspring_self<..., 0, GRP> then finds its prefix through them
  const bool grp_on = (grp_arg & 0x100) != 0;
  const int grp_set = (grp_arg >> 9) & 3, grp_shift = 5 + ((grp_arg >> 11) & 3);
  // all5 != nullptr (sharded series whose levels come from the global max, collectives issued by the library): the max is the
  // largest of the ranks' order keys (word 4 of each rank's 5 all-gathered words) -- every block decodes it itself, block 0
  // publishes it with the level it selects (k_boundary_pack and the status read find them in Scalars)
  // hold_mask (sharded series, level from the all-gathered max): while the series is on hold after a capacity miss (err bit 3)
  // or void (bit 2) the sums of the observation it holds at must survive the observations enqueued behind it -- the host
  // resumes exactly there (cssm_pf_shard_resume)
  if (hold_mask && (sc->err & hold_mask)) return;
  __shared__ cssm_u128 s_a[CSSM_BLOCK / 64], s_b[CSSM_BLOCK / 64];
  const double* tab = nullptr; (void)logtab;   // literal constants measured faster than an LDS constant table (LABNOTES.md)
  double pre[CSSM_ITEMS];   // the block's first tile is requested before the (serial) max decode
  if (blockIdx.x < nunits) load_tile_raw(logw, (uint64_t)blockIdx.x * sup * CSSM_TILE, n, raw, pre);
  // slot_set < 0: the max was agreed elsewhere (sharded: all-reduced value at gmax_in; stateless: sc->gmax)
  double gmax_dec;
  if (all5) {
    unsigned long long key = 0ull;
    for (int r = 0; r < world; ++r) { const unsigned long long k = all5[5 * r + 4]; key = (k > key) ? k : key; }
    gmax_dec = cssm_order_unkey(key);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->gmax = gmax_dec; sc->ref = cssm_ref_choose(rec->ref, gmax_dec); }
  } else {
    gmax_dec = (slot_set >= 0) ? block_decode_slots(sc, slot_set) : (gmax_in ? *gmax_in : sc->gmax);
  }
  // the level every kernel of the step agrees on: the observation's reference level when the max allows it
  const double gmax = raw ? gmax_dec : cssm_ref_choose(rec->ref, gmax_dec);
  for (uint32_t unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const uint32_t t0 = unit * sup;
    const uint32_t t1 = (t0 + sup < ntiles) ? t0 + sup : ntiles;
    cssm_u128 a = cssm_u128_zero(), b = cssm_u128_zero();
    for (uint32_t tile = t0; tile < t1; ++tile) {
      double w1[CSSM_ITEMS];
      if (unit == blockIdx.x && tile == t0) weights_from_raw(pre, gmax, raw, w1, tab);
      else load_tile_weights(logw, (uint64_t)tile * CSSM_TILE, n, gmax, raw, w1, tab);
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) {
        if (raw) {   // (the stateless entry point: arbitrary host doubles, range-checked)
          a = cssm_u128_add(a, cssm_fix_from_double(w1[r]));
          b = cssm_u128_add(b, cssm_fix_from_double(w1[r] * w1[r]));
        } else {
          a = cssm_u128_add(a, cssm_fix_from_unit(w1[r]));
          b = cssm_u128_add(b, cssm_fix_from_unit(w1[r] * w1[r]));
        }
      }
    }
    a = wave_sum_u128(a);
    b = wave_sum_u128(b);
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      cssm_u128 ta = s_a[0], tb = s_b[0];
#pragma unroll
      for (int w = 1; w < CSSM_BLOCK / 64; ++w) { ta = cssm_u128_add(ta, s_a[w]); tb = cssm_u128_add(tb, s_b[w]); }
      unitS[unit] = ta; unitS2[unit] = tb;
      if (grp_on) group_sums_add(sc, grp_set, unit >> grp_shift, ta, tb, false);
    }
    __syncthreads();
  }
}

// Large clouds run k_propagate's single-tile kernel -- many blocks per unit of sums -- and this kernel folds the blocks' sums
// into the units' (one WAVE per unit: lane l adds the entries l, l + 64, ... of its unit; integer sums: any grouping gives the
// same bits), so that k_offspring reads <= 1024 unit sums whatever the cloud's size.  ~5 us per weighted observation; what the
// single-tile kernel saves grows with the latent dimension (cssm_pf.hip, uses_fine: d >= 4).
static __global__ __launch_bounds__(CSSM_BLOCK) void k_reduce_units(const cssm_u128* __restrict__ blockS, const cssm_u128* __restrict__ blockS2,
                                                             uint32_t nblocks, uint32_t blocks_per_unit, uint32_t nunits,
                                                             cssm_u128* __restrict__ unitS, cssm_u128* __restrict__ unitS2,
                                                             const Scalars* __restrict__ sc, const StepRec* __restrict__ rec) {
  if (!rec->has_obs || (sc->err & (4u | 8u | 16u | 64u))) return;   // (nothing was formed: an unweighted observation, a series on hold)
  const uint32_t unit = blockIdx.x * (CSSM_BLOCK / 64) + (threadIdx.x >> 6);
  if (unit >= nunits) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t b0 = unit * blocks_per_unit;
  uint32_t b1 = b0 + blocks_per_unit;
  b1 = (b1 < nblocks) ? b1 : nblocks;
  cssm_u128 a = cssm_u128_zero(), b = cssm_u128_zero();
  // blockS2 == nullptr: the blocks formed no sums of squares (single GPU: k_offspring forms them)
  for (uint32_t q = b0 + lane; q < b1; q += 64u) { a = cssm_u128_add(a, blockS[q]); if (blockS2) b = cssm_u128_add(b, blockS2[q]); }
  a = wave_sum_u128(a);
  if (blockS2) b = wave_sum_u128(b);
  if (lane == 0u) { unitS[unit] = a; if (blockS2) unitS2[unit] = b; }
}

// Exclusive scan of the tile sums in one block (thread t owns a contiguous chunk of tiles: sum, block
// scan of the 1024 chunk sums, then prefix write-back); local totals; with `single` also ll / ess.
static __global__ __launch_bounds__(1024) void k_scan_tiles(const cssm_u128* __restrict__ tileS, const cssm_u128* __restrict__ tileS2,
                                                     cssm_u128* __restrict__ tileP, uint32_t ntiles, Scalars* sc,
                                                     uint64_t n_global, int single,
                                                     double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t rec_idx,
                                                     const double* __restrict__ gmax_in, unsigned long long* __restrict__ sums4_out,
                                                     int export_max, uint32_t hold_mask, int sums_valid) {
  // hold_mask: see k_tile_sums (neither the exported words nor the max slots may change while a sharded series is on hold);
  // sums_valid = 0: only the max travels (an LGCP observation before its level is known) -- the sums read as zero
  if (hold_mask && (sc->err & hold_mask)) return;
  __shared__ cssm_u128 s_w[16], s_w2[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (gmax_in && threadIdx.x == 0) { sc->gmax = *gmax_in; sc->ref = *gmax_in; }
  if (export_max && threadIdx.x < 64) {   // sharded: word 4 = order key of the local max (slot set 0), slots cleared for the next step
    unsigned long long k = 0ull;
    if (threadIdx.x < CSSM_MAXSLOTS) {
      k = sc->maxslot[(size_t)threadIdx.x * CSSM_SLOT_STRIDE];
      sc->maxslot[(size_t)threadIdx.x * CSSM_SLOT_STRIDE] = 0ull;
    }
    k = wave_max_u64(k);
    if (threadIdx.x == 0) sums4_out[4] = k;
  }
  const uint32_t chunk = (ntiles + 1023u) / 1024u;
  const uint32_t t0 = threadIdx.x * chunk;
  const uint32_t t1 = (t0 + chunk < ntiles) ? t0 + chunk : ntiles;
  cssm_u128 v = cssm_u128_zero(), v2 = cssm_u128_zero();
  if (sums_valid) for (uint32_t t = t0; t < t1; ++t) { v = cssm_u128_add(v, tileS[t]); v2 = cssm_u128_add(v2, tileS2[t]); }
  cssm_u128 inc = wave_scan_u128(v, lane);
  cssm_u128 tot2 = wave_sum_u128(v2);
  if (lane == 63) s_w[wid] = inc;
  if (lane == 0) s_w2[wid] = tot2;
  __syncthreads();
  cssm_u128 off = cssm_u128_zero();
  for (int w = 0; w < wid; ++w) off = cssm_u128_add(off, s_w[w]);
  // exclusive prefix of this thread's chunk = off + inc - v (integers: exact)
  cssm_u128 run = cssm_u128_add(off, inc);
  { cssm_u128 r; r.lo = run.lo - v.lo; r.hi = run.hi - v.hi - (run.lo < v.lo ? 1u : 0u); run = r; }
  if (sums_valid) for (uint32_t t = t0; t < t1; ++t) { tileP[t] = run; run = cssm_u128_add(run, tileS[t]); }
  if (threadIdx.x == 0) {
    cssm_u128 c = cssm_u128_zero(), c2 = cssm_u128_zero();
    for (int w = 0; w < 16; ++w) { c = cssm_u128_add(c, s_w[w]); c2 = cssm_u128_add(c2, s_w2[w]); }
    sc->S_local = c; sc->S2_local = c2;
    if (sums4_out) { sums4_out[0] = c.lo; sums4_out[1] = c.hi; sums4_out[2] = c2.lo; sums4_out[3] = c2.hi; }
    if (single) {
      sc->S_off = cssm_u128_zero();
      sc->S_tot = c; sc->S2_tot = c2;
      finish_step(sc, n_global);
      if (ll_t) { ll_t[rec_idx] = sc->ll; ess_t[rec_idx] = sc->ess; }
    }
  }
}

// ------------------------------------------------------------------------------------ offspring (end slots): cssm_offspring.hip.h
#include "cssm_offspring.hip.h"

// Resampling.multinomialResampling (model/Resampling.scala:92-96): slot i draws its own uniform and takes the
// first particle whose cumulative normalised weight reaches it (breeze Multinomial.draw); the output is in
// draw order, not sorted.  `cum` is non-decreasing and ends at exactly 1.0.
static __global__ void k_multinomial(const double* __restrict__ cum, uint64_t n, uint64_t seed, uint32_t step, uint32_t* __restrict__ anc) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const double ui = cssm_multi_uniform(seed, step, i);
    uint64_t lo = 0, hi = n - 1;      // first j with cum[j] >= ui (exists: cum[n-1] == 1 > ui)
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (cum[mid] >= ui) hi = mid; else lo = mid + 1;
    }
    anc[i] = (uint32_t)lo;
  }
}

// ------------------------------------------------------------------------------------ gather / pick

// PfState.particles on demand: out[k][i] = src[k][anc[i]]
static __global__ void k_gather(const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc,
                         double* __restrict__ out, size_t out_stride, uint64_t n, int d,
                         const double* __restrict__ src2, size_t src2_stride, uint32_t n_split) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const size_t j = anc ? (size_t)anc[i] : (size_t)i;
    for (int k = 0; k < d; ++k)
      out[(size_t)k * out_stride + i] = (src2 && j >= n_split)   // (rows received from other ranks: system-scope loads, as everywhere a window is read)
          ? (src2_stride == 0 ? ld_sys_f64(src2 + (size_t)(j - n_split) * (size_t)(d + 1) + k) : ld_sys_f64(src2 + (size_t)k * src2_stride + (j - n_split)))
          : src[(size_t)k * src_stride + j];
  }
}

// Resampling.sampleOne for `filter` (model/ParticleFilter.scala:157): one particle of the current cloud
static __global__ void k_pick(const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc,
                       uint64_t idx, int d, double* __restrict__ out_row) {
  const int k = threadIdx.x;
  if (k < d) {
    const size_t j = anc ? (size_t)anc[idx] : (size_t)idx;
    out_row[k] = src[(size_t)k * src_stride + j];
  }
}

// The end of a call, instead of device-to-host copies: (1) an ESS still pending (Scalars::pend: the last weighted observation's
// blocks left partial sums of squared weights, nobody after it totalled them) is formed and filed; (2) the call's per-observation
// results and the scalars the host reads (Scalars from `err` on) are written straight into host-mapped memory.  One block; the
// host only synchronises the stream.  (Four small hipMemcpyAsync D2H behind a K = 20 series cost ~45 us, this kernel ~4.)
static __global__ __launch_bounds__(CSSM_BLOCK) void k_finish(Scalars* __restrict__ sc, const cssm_u128* __restrict__ s2buf, uint32_t s2_stride,
                                                              double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t T, uint32_t gen,
                                                              Scalars* __restrict__ host_sc, double* __restrict__ host_ll_t, int32_t* __restrict__ host_ess_t,
                                                              uint32_t* __restrict__ host_done = nullptr, uint32_t done_seq = 0u) {
  __shared__ cssm_u128 s_red[CSSM_BLOCK / 64];
  __shared__ int32_t s_ess;
  const uint32_t pend = sc->pend, p_buf = sc->pend_buf, p_n = sc->pend_n, p_idx = sc->pend_idx, p_gen = sc->pend_gen, err = sc->err;
  const cssm_u128 p_S = sc->pend_S;
  int32_t ess = sc->ess;
  if (pend && p_buf < 2u && p_n <= s2_stride) {           // (uniform)
    cssm_u128 t2 = cssm_u128_zero();
    const cssm_u128* pb = s2buf + (size_t)p_buf * s2_stride;
    for (uint32_t q = threadIdx.x; q < p_n; q += CSSM_BLOCK) t2 = cssm_u128_add(t2, pb[q]);
    t2 = block_sum_u128(t2, s_red);
    if (threadIdx.x == 0) {
      if (!(err & 3u) && !cssm_u128_is_zero(p_S)) ess = cssm_ess_of(p_S, t2);
      sc->ess = ess; sc->pend = 0u;
      if (ess_t && p_gen == gen && p_idx < T) ess_t[p_idx] = ess;
      s_ess = ess;
    }
    __syncthreads();
    ess = s_ess;
  }
  // per-observation results of the call (device arrays, written by the publisher blocks / k_record) -> the host's mirrors
  if (host_ll_t) for (uint32_t s = threadIdx.x; s < T; s += CSSM_BLOCK) host_ll_t[s] = ll_t[s];
  if (host_ess_t) for (uint32_t s = threadIdx.x; s < T; s += CSSM_BLOCK) host_ess_t[s] = (pend && p_gen == gen && s == p_idx) ? ess : ess_t[s];
  // the scalars from `err` on, word for word; ess and pend as just settled
  constexpr uint32_t W0 = (uint32_t)(offsetof(Scalars, err) / 4), W1 = (uint32_t)(sizeof(Scalars) / 4);
  constexpr uint32_t WE = (uint32_t)(offsetof(Scalars, ess) / 4), WP = (uint32_t)(offsetof(Scalars, pend) / 4);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(sc);
  uint32_t* dst = reinterpret_cast<uint32_t*>(host_sc);
  constexpr uint32_t WT = (uint32_t)(offsetof(Scalars, t_last) / 4);
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();   // (Scalars::t_last: with t_first the device time of the call)
  for (uint32_t w = W0 + threadIdx.x; w < W1; w += CSSM_BLOCK)
    dst[w] = (w == WE) ? (uint32_t)ess : ((w == WP) ? 0u : ((w == WT) ? (uint32_t)now : ((w == WT + 1u) ? (uint32_t)(now >> 32) : src[w])));
  // the call's completion word, behind everything above: the host polls it instead of waiting for the stream (read_scalars)
  if (host_done) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_done, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The first record of a call that CONTINUES a filter whose levels are predicted (LGCP): its level is what the last weighted
// observation before the call published (Scalars::next_ref) -- on the stream, the host never learns it.
static __global__ void k_chain_level(StepRec* __restrict__ rec, const Scalars* __restrict__ sc) { rec->ref = sc->next_ref; }

// per-step record of results for the batch drivers
static __global__ void k_record(const Scalars* __restrict__ sc, double* __restrict__ ll_t, int32_t* __restrict__ ess_t, uint32_t s) {
  ll_t[s] = sc->ll;
  ess_t[s] = sc->ess;
}

// ------------------------------------------------------------------------------------ cloud summaries
// getIntervals (model/ParticleFilter.scala:415-424): meanState (:465-479), getallCredibleIntervals
// (:488-512) and getOrderStatistic (:455-460) of the CURRENT (resampled) cloud, on the device.

// link of the observing leaf (model/Model.scala:24,183,269,296,318-326,345)
__device__ __forceinline__ double link_of(int obs_kind, double g) {
  switch (obs_kind) {
    case CSSM_OBS_POISSON: case CSSM_OBS_NEGBIN: case CSSM_OBS_ZIP: return cssm_exp(g);
    case CSSM_OBS_BERNOULLI: return (g > 6.0) ? 1.0 : ((g < -6.0) ? 0.0 : 1.0 / (1.0 + cssm_exp(-g)));
    case CSSM_OBS_BETA: return cssm_exp(-g);
    default: return g;
  }
}

// rows 0..d-1: the resampled state components, row d: eta = link(f(x, t)); stored as order-preserving keys
template <int D>
__global__ __launch_bounds__(CSSM_BLOCK) void k_summary_fill(const double* __restrict__ src, size_t src_stride,
                                                             const uint32_t* __restrict__ anc, const double* __restrict__ src2,
                                                             size_t src2_stride, uint32_t n_split, uint64_t n,
                                                             const StepRec* __restrict__ rec, ModelK mk,
                                                             unsigned long long* __restrict__ keys, size_t kstride,
                                                             double* __restrict__ partial /*[gridDim.x][D]*/) {
  __shared__ double s_p[CSSM_BLOCK / 64][D];
  double acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CSSM_BLOCK) {
    const size_t j = anc ? (size_t)anc[i] : (size_t)i;
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      x[k] = (src2 && j >= n_split)
          ? (src2_stride == 0 ? ld_sys_f64(src2 + (size_t)(j - n_split) * (size_t)(D + 1) + k) : ld_sys_f64(src2 + (size_t)k * src2_stride + (j - n_split)))
          : src[(size_t)k * src_stride + j];
      keys[(size_t)k * kstride + i] = cssm_order_key(x[k]);
      acc[k] += x[k];
    }
    keys[(size_t)D * kstride + i] = cssm_order_key(link_of(mk.obs_kind, gamma_of<D>(mk, rec, x)));
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    double v = 0.0;
    for (int w = 0; w < CSSM_BLOCK / 64; ++w) v += s_p[w][threadIdx.x];
    partial[(size_t)blockIdx.x * D + threadIdx.x] = v;
  }
}

// Radix select, most significant byte first, two targets (lower / upper order statistic) per row.
struct SelState { unsigned long long prefix[2]; unsigned long long rank[2]; };

static __global__ __launch_bounds__(CSSM_BLOCK) void k_sel_hist(const unsigned long long* __restrict__ keys, size_t kstride, uint64_t n,
                                                         const SelState* __restrict__ st, int shift, uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_h[2][256];
  const int row = blockIdx.y;
  for (int i = threadIdx.x; i < 512; i += CSSM_BLOCK) (&s_h[0][0])[i] = 0;
  __syncthreads();
  const unsigned long long p0 = st[row].prefix[0], p1 = st[row].prefix[1];
  const unsigned long long hm = (shift >= 56) ? 0ull : (~0ull << (shift + 8));   // bits above the current byte
  const unsigned long long* kr = keys + (size_t)row * kstride;
  for (uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CSSM_BLOCK) {
    const unsigned long long k = kr[i];
    const uint32_t b = (uint32_t)(k >> shift) & 255u;
    if ((k & hm) == (p0 & hm)) atomicAdd(&s_h[0][b], 1u);
    if ((k & hm) == (p1 & hm)) atomicAdd(&s_h[1][b], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += CSSM_BLOCK) {
    const uint32_t v = (&s_h[0][0])[i];
    if (v) atomicAdd(&hist[(size_t)row * 512 + i], v);
  }
}

static __global__ void k_sel_pick(SelState* __restrict__ st, int shift, uint32_t* __restrict__ hist) {   // <<<rows, 2>>>
  const int row = blockIdx.x, t = threadIdx.x;
  uint32_t* h = hist + (size_t)row * 512 + t * 256;
  unsigned long long r = st[row].rank[t], cum = 0;
  int b = 0;
  for (; b < 255; ++b) {
    if (cum + h[b] > r) break;
    cum += h[b];
  }
  st[row].prefix[t] |= (unsigned long long)b << shift;
  st[row].rank[t] = r - cum;
  for (int i = 0; i < 256; ++i) h[i] = 0;
}

static __global__ void k_summary_finish(const SelState* __restrict__ st, const double* __restrict__ partial, int nblocks, int d, uint64_t n,
                                 double* __restrict__ out /*[3][d+1]: mean, lower, upper*/) {
  const int k = threadIdx.x;
  if (k > d) return;
  if (k < d) {
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * d + k];
    out[k] = s / (double)n;
  } else {
    out[k] = 0.0;   // eta of the mean state is formed on the host: link(f(mean, t))
  }
  out[(d + 1) + k] = cssm_order_unkey(st[k].prefix[0]);
  out[2 * (d + 1) + k] = cssm_order_unkey(st[k].prefix[1]);
}

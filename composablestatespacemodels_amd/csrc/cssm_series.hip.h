// cssm_series.hip.h -- the persistent series kernel: ONE cooperative launch runs all T observations of llFilter / filter
// (model/ParticleFilter.scala:137-140,152-158) on one GPU.
//
// Why: a dependent kernel launch costs ~5-10 us on MI355X however little it does, so a filter step made of launches has a
// floor of ~20 us; at N = 100 000 (PMMH) that floor is the whole step and at N = 2^20 half of it.
// Here the grid stays resident; per weighted observation it runs
//
//   phase P   propagate_range (the body of k_propagate): gather through the previous ancestors, transition, f, log-density;
//             the block keeps its particles' log-weights IN LDS and forms its fixed-point sums of exp(w - c) and its max
//   exchange  grid barrier that also reduces: every block learns the sum of the blocks before it, the totals and the max
//   phase O   (same block, same particles) cumulative weights -> end slots -> the ancestor indices of the slots its
//             particles own, assembled per 2048-slot chunk in LDS and written out as whole lines
//   barrier   before the next observation gathers through those indices
//
// so the log-weights never travel through HBM, exp(w - c) is not evaluated in a second kernel, and a step boundary costs
// a barrier (~3 us) instead of a launch.  Inside a kernel the eight XCDs' L2 caches are not coherent with each other, so
// every datum that crosses blocks -- state rows, ancestor indices, the exchange records -- is stored write-through and
// loaded with agent scope (sc1); the barrier itself is fence-free (tools/barrier_bench.hip: a __threadfence() per block
// costs 40-110 ns EACH because every fence is an L2 write-back + invalidate queued behind the others of its XCD).
//
// Exit condition: every block passes every barrier of the series exactly once; a spin that does not end within
// CSSM_SER_SPIN_LIMIT polls raises the abort bit, which every other spin sees, sets err bit 4 (16) and ends the kernel.
// The launch is cooperative (hipLaunchCooperativeKernel), so all blocks are resident before any of them spins.
//
// Results are bit-identical to the per-observation kernels (k_propagate + k_tile_sums + k_offspring): the per-particle
// functions are the same code, and every sum is an integer sum (tests/test_gpu_parity.py runs both paths).
#pragma once

#include "cssm_propagate.hip.h"
#include "cssm_series_abi.h"

template <int D> struct SeriesItems { static constexpr int value = (D <= 8) ? 2 : 1; };   // particles per thread in phase P


// what a block contributes to / learns from the exchange
struct SerRec { unsigned long long S_lo, S_hi, S2_lo, S2_hi, key, pad_[3]; };   // 64 bytes
static_assert(sizeof(SerRec) == 64, "one record per half line");

// Device memory of the barrier, zeroed by the host before every launch.  Counters only grow (no reset races): round r of
// the barrier is complete when a group counter reaches members * (r + 1) and the top counter groups * (r + 1).
struct SeriesSync {
  unsigned long long flag; unsigned long long pad0_[15];          // last completed round + 1; bit 63 = abort
  unsigned top; unsigned pad1_[31];
  unsigned gcnt[CSSM_SER_MAXBLOCKS / CSSM_SER_GROUP][32];          // one 128-byte line per group (word 0)
  SerRec grp[2][CSSM_SER_MAXBLOCKS / CSSM_SER_GROUP];              // group totals, by exchange parity
  SerRec rec[2][CSSM_SER_MAXBLOCKS];                               // block records, by exchange parity
};

__device__ __forceinline__ unsigned long long ser_ld(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1
}
__device__ __forceinline__ void ser_st(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ void ser_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// what the exchange hands every thread of the block
struct SerSums { cssm_u128 pre, tot, tot2; unsigned long long key; };

// Grid barrier, optionally with the reduction of one record per block (WITH_DATA).  All threads of all blocks call it the
// same number of times with the same template argument.  Returns false when the barrier was aborted (the caller leaves
// the kernel).  `round` counts barriers, `xpar` exchanges (buffer parity of rec / grp: a buffer is rewritten two exchanges
// later, when every block has long passed the barrier behind its last read).
template <bool WITH_DATA>
__device__ __forceinline__ bool ser_sync(SeriesSync* __restrict__ sy, unsigned& round, unsigned& xpar, cssm_u128 S, cssm_u128 S2,
                                         unsigned long long key, SerSums* out) {
  __shared__ SerSums s_out;
  __shared__ int s_ok;
  // everything this block stored (write-through) must have left the CU before the block arrives: s_barrier alone does
  // not wait for outstanding stores
  ser_wait_vm();
  __syncthreads();
  const unsigned nb = gridDim.x, b = blockIdx.x, g = b / CSSM_SER_GROUP, ng = (nb + CSSM_SER_GROUP - 1) / CSSM_SER_GROUP;
  const unsigned in_g = (g + 1 == ng) ? nb - g * CSSM_SER_GROUP : CSSM_SER_GROUP;
  const unsigned par = xpar & 1u;
  if (threadIdx.x < 64) {                        // wave 0 does the talking (uniform control flow inside)
    const unsigned lane = threadIdx.x;
    if (WITH_DATA && lane == 0) {
      SerRec* r = &sy->rec[par][b];
      ser_st(&r->S_lo, S.lo); ser_st(&r->S_hi, S.hi); ser_st(&r->S2_lo, S2.lo); ser_st(&r->S2_hi, S2.hi); ser_st(&r->key, key);
    }
    ser_wait_vm();
    unsigned a = 0;
    if (lane == 0) a = __hip_atomic_fetch_add(&sy->gcnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    if (a == in_g * (round + 1u) - 1u) {         // last block of its group: the group's total, then the top counter
      if (WITH_DATA) {
        cssm_u128 gs = cssm_u128_zero(), gs2 = cssm_u128_zero();
        unsigned long long gk = 0ull;
        if (lane < in_g) {
          const SerRec* r = &sy->rec[par][g * CSSM_SER_GROUP + lane];
          gs.lo = ser_ld(&r->S_lo); gs.hi = ser_ld(&r->S_hi); gs2.lo = ser_ld(&r->S2_lo); gs2.hi = ser_ld(&r->S2_hi); gk = ser_ld(&r->key);
        }
        gs = wave_sum_u128(gs); gs2 = wave_sum_u128(gs2); gk = wave_max_u64(gk);
        if (lane == 0) {
          SerRec* r = &sy->grp[par][g];
          ser_st(&r->S_lo, gs.lo); ser_st(&r->S_hi, gs.hi); ser_st(&r->S2_lo, gs2.lo); ser_st(&r->S2_hi, gs2.hi); ser_st(&r->key, gk);
        }
        ser_wait_vm();
      }
      unsigned t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(&sy->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      if (t == ng * (round + 1u) - 1u && lane == 0) {
        // (an abort raised meanwhile must survive: fetch_max keeps bit 63)
        __hip_atomic_fetch_max(&sy->flag, (unsigned long long)(round + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // wait for the round to be published (all lanes poll the same word: one request)
    unsigned spins = 0;
    unsigned long long f = ser_ld(&sy->flag);
    while ((f & ~CSSM_SER_ABORT) < (unsigned long long)(round + 1u) && !(f & CSSM_SER_ABORT)) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CSSM_SER_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_fetch_or(&sy->flag, CSSM_SER_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = CSSM_SER_ABORT;
        break;
      }
      f = ser_ld(&sy->flag);
    }
    const bool ok = !(f & CSSM_SER_ABORT);
    if (WITH_DATA && ok) {
      // lanes 0..31: one group total each (prefix over the groups before mine, grand totals, max);
      // lanes 32..63: the records of the blocks of my group before me
      cssm_u128 v = cssm_u128_zero(), v2 = cssm_u128_zero();
      unsigned long long k = 0ull;
      bool in_pre = false;
      if (lane < 32u) {
        if (lane < ng) {
          const SerRec* r = &sy->grp[par][lane];
          v.lo = ser_ld(&r->S_lo); v.hi = ser_ld(&r->S_hi); v2.lo = ser_ld(&r->S2_lo); v2.hi = ser_ld(&r->S2_hi); k = ser_ld(&r->key);
          in_pre = lane < g;
        }
      } else if (lane - 32u < b - g * CSSM_SER_GROUP) {
        const SerRec* r = &sy->rec[par][g * CSSM_SER_GROUP + (lane - 32u)];
        v.lo = ser_ld(&r->S_lo); v.hi = ser_ld(&r->S_hi);
        in_pre = true;
      }
      const cssm_u128 zero = cssm_u128_zero();
      const cssm_u128 pre = wave_sum_u128(in_pre ? v : zero);
      const cssm_u128 tot = wave_sum_u128(lane < 32u ? v : zero);
      const cssm_u128 tot2 = wave_sum_u128(lane < 32u ? v2 : zero);
      const unsigned long long kk = wave_max_u64(k);
      if (lane == 0) { s_out.pre = pre; s_out.tot = tot; s_out.tot2 = tot2; s_out.key = kk; }
    }
    if (lane == 0) s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  round += 1u;
  if (WITH_DATA) { xpar += 1u; if (out) *out = s_out; }
  const bool ok = s_ok != 0;
  __syncthreads();   // s_out / s_ok may be rewritten by the next call
  return ok;
}

// Phase O of one weighted observation for the block's particles [range_lo, range_lo + cnt): log-weights in s_lw.
// treeEcdf + findAllInTreeMap of model/Resampling.scala:36-58,69 exactly as k_offspring computes them (same contract
// functions, same fast path for the end slot); the ancestor indices of the slots the block's particles own are assembled
// in LDS -- every particle drops its index + 1 at the first slot of its run, an inclusive max-scan fills the runs (indices
// grow with the slots) -- and leave as whole 32-byte pieces per thread, write-through.
// s_slot: CSSM_RUN_CHUNK words of LDS.
__device__ __forceinline__ void series_offspring(const double* __restrict__ s_lw, uint32_t range_lo, uint32_t cnt, double level,
                                                 cssm_u128 pre, cssm_u128 tot, double u, uint64_t n_global, uint32_t* __restrict__ anc,
                                                 int force_exact, uint32_t* __restrict__ s_slot) {
  __shared__ cssm_u128 s_w[CSSM_BLOCK / 64];
  __shared__ uint32_t s_last[CSSM_BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool pow2 = (n_global & (n_global - 1)) == 0;
  const double inv_n = 1.0 / (double)n_global;
  const double totd = cssm_u128_to_double(tot);
  const double nd = (double)n_global;
  const double scale = nd / totd;
  const double eps = nd * 0x1.0p-46;
  const uint32_t n_last = (uint32_t)(n_global - 1);
  auto exact_count = [&](cssm_u128 G) -> uint32_t {
    const double C = cssm_u128_to_double(G) / totd;
    return (uint32_t)(pow2 ? cssm_sys_count_pow2(C, u, n_global, inv_n) : cssm_sys_count(C, u, n_global));
  };
  cssm_u128 toff = pre;                                   // cumulative weight before the current tile
  uint32_t tile_b = (range_lo == 0) ? 0u : exact_count(pre);   // first slot of the tile's first particle (uniform)
  for (uint32_t base = 0; base < cnt; base += CSSM_TILE) {
    // ---- cumulative weights and end slots of the tile's particles (4 per thread), as in k_offspring
    const uint32_t p0 = base + threadIdx.x * CSSM_ITEMS;
    cssm_u128 q[CSSM_ITEMS];
    cssm_u128 tsum = cssm_u128_zero();
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      const double w1 = (p0 + r < cnt) ? cssm_exp(s_lw[p0 + r] - level) : 0.0;
      q[r] = cssm_fix_from_double(w1);
      tsum = cssm_u128_add(tsum, q[r]);
    }
    const cssm_u128 inc = wave_scan_u128(tsum, lane);
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    cssm_u128 off = toff;
    for (int w = 0; w < wid; ++w) off = cssm_u128_add(off, s_w[w]);
    cssm_u128 run = cssm_u128_add(off, inc);
    { cssm_u128 t; t.lo = run.lo - tsum.lo; t.hi = run.hi - tsum.hi - (run.lo < tsum.lo ? 1u : 0u); run = t; }
    uint32_t e[CSSM_ITEMS];
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      run = cssm_u128_add(run, q[r]);
      const double sd = cssm_fma((double)run.hi, 0x1.0p64, (double)run.lo);
      const double pp = cssm_fma(sd, scale, -u);
      const double fl = __builtin_floor(pp);
      const double fr = pp - fl;
      double c = fl + 1.0;
      c = (c < 0.0) ? 0.0 : c;
      c = (c > nd) ? nd : c;
      const bool safe = (fr > eps) && (fr < 1.0 - eps) && !force_exact;
      e[r] = safe ? (uint32_t)c : exact_count(run);
    }
    if (lane == 63) s_last[wid] = e[CSSM_ITEMS - 1];
    uint32_t prev = dpp0<0x138 /* wave_shr:1 */, 0xf>(e[CSSM_ITEMS - 1]);
    __syncthreads();
    if (lane == 0) prev = (wid > 0) ? s_last[wid - 1] : tile_b;
    uint32_t tile_e = s_last[CSSM_BLOCK / 64 - 1];        // end slot of the tile's last particle (padding items repeat it)
    // defensive bounds (a NaN total or a broken invariant must never become a wild store): slots live in [0, N]
    tile_e = (tile_e > (uint32_t)n_global) ? (uint32_t)n_global : tile_e;
    tile_b = (tile_b > tile_e) ? tile_e : tile_b;
    {   // this wave's slot range, clipped to the tile's (defensive) bounds
      uint32_t wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
      uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)e[CSSM_ITEMS - 1], 63);
      wb = (wb < tile_b) ? tile_b : wb; we = (we > tile_e) ? tile_e : we; wb = (wb > we) ? we : wb;
      fill_runs_wave<true>(prev, e, range_lo + p0, wb, we, anc, 0u, n_last, s_slot + wid * CSSM_WAVE_CHUNK);
    }
    // ---- next tile
    cssm_u128 ttot = s_w[0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) ttot = cssm_u128_add(ttot, s_w[w]);
    toff = cssm_u128_add(toff, ttot);
    tile_b = tile_e;
    __syncthreads();
  }
}

// block-wide totals of the thread-local fixed-point sums and of the max (uniform results)
__device__ __forceinline__ void series_block_reduce(cssm_u128& S, cssm_u128& S2, double& tmax, bool& bad) {
  __shared__ cssm_u128 s_a[CSSM_BLOCK / 64], s_b[CSSM_BLOCK / 64];
  __shared__ unsigned long long s_k[CSSM_BLOCK / 64];
  __shared__ int s_bad;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  S = wave_sum_u128(S); S2 = wave_sum_u128(S2);
  const unsigned long long k = wave_max_u64(cssm_order_key(tmax));
  const bool any_bad = __any(bad);
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  if (lane == 0) { s_a[wid] = S; s_b[wid] = S2; s_k[wid] = k; if (any_bad) s_bad = 1; }
  __syncthreads();
  S = s_a[0]; S2 = s_b[0];
  unsigned long long kk = s_k[0];
#pragma unroll
  for (int w = 1; w < CSSM_BLOCK / 64; ++w) { S = cssm_u128_add(S, s_a[w]); S2 = cssm_u128_add(S2, s_b[w]); kk = (s_k[w] > kk) ? s_k[w] : kk; }
  tmax = cssm_order_unkey(kk);
  bad = s_bad != 0;
  __syncthreads();
}

#ifndef CSSM_SER_WAVES
#define CSSM_SER_WAVES 4
#endif

template <int D, int OBS>
__global__ __launch_bounds__(CSSM_BLOCK, CSSM_SER_WAVES) void k_series(
    double* __restrict__ state0, double* __restrict__ state1, size_t stride, uint32_t* __restrict__ anc, double* __restrict__ logw,
    uint64_t n_arg, uint64_t seed, const StepRec* __restrict__ recs, uint32_t T, ModelK mk, Scalars* __restrict__ sc,
    SeriesSync* __restrict__ sy, const double* __restrict__ logtab, uint32_t per_block, int cur0, int force_exact,
    double* __restrict__ ll_t, int32_t* __restrict__ ess_t, double* __restrict__ path, unsigned long long* __restrict__ ts_arg,
    uint32_t ts_blocks) {
  constexpr int IT = SeriesItems<D>::value;
  constexpr int STAGE_BYTES = PropStage<D, IT>::bytes;
  constexpr int BUF_BYTES = STAGE_BYTES > CSSM_RUN_CHUNK * 4 ? STAGE_BYTES : CSSM_RUN_CHUNK * 4;
  __shared__ __attribute__((aligned(16))) unsigned char s_buf[BUF_BYTES];   // phase P: LDS staging; phase O: slot chunk
  extern __shared__ __attribute__((aligned(16))) double s_lw[];             // per_block log-weights
  const double* tab = stage_log_table(logtab);
  const uint32_t n = (uint32_t)n_arg;
  const uint32_t range_lo = blockIdx.x * per_block;
  const uint32_t range_hi = (range_lo + per_block < n) ? range_lo + per_block : n;   // (the host launches no empty block)
  const uint32_t cnt = range_hi - range_lo;
  unsigned round = 0, xpar = 0;
  int cur = cur0;                 // state buffer holding the cloud the next observation reads
  bool via_anc = false;           // ... through the ancestor indices of the previous observation's resampling
  // profiling: thread 0 of the first ts_blocks blocks stamps its own row of T x 5 ticks (block 0 alone: the phase averages)
  const bool stamp = (ts_arg != nullptr) && blockIdx.x < ts_blocks && threadIdx.x == 0;
  unsigned long long* const ts = ts_arg + (size_t)blockIdx.x * T * CSSM_SER_TS_PER_STEP;
  for (uint32_t s = 0; s < T; ++s) {
    const StepRec* rec = recs + s;
    const int has_obs = rec->has_obs;
    const double* src = cur ? state1 : state0;
    double* dst = cur ? state0 : state1;
    if (stamp) ts[(size_t)s * CSSM_SER_TS_PER_STEP + 0] = wall_clock64();
    // ---- phase P
    PropAcc acc;
    double* pick_out = (path != nullptr && s >= 1) ? path + (size_t)s * D : nullptr;
    // (logw: only the last observation's log-weights go to memory, for cssm_pf_get_logw)
    propagate_range<D, false, IT, OBS, true, true>(src, stride, via_anc ? anc : nullptr, dst, stride, (s + 1 == T) ? logw : nullptr, 0ull, seed,
                                                   rec, mk, nullptr, 0, 0u, tab, range_lo, range_hi, 1, pick_out,
                                                   s >= 1 ? recs[s - 1].pick : 0u, s_lw, s_buf, acc, n - 1u, &sc->err);
    cur ^= 1;
    if (!has_obs) {               // model/ParticleFilter.scala:121: propagated cloud, ll and ess unchanged
      if (blockIdx.x == 0 && threadIdx.x == 0 && ll_t) { ll_t[s] = sc->ll; ess_t[s] = sc->ess; }
      if (via_anc) {
        // other blocks may still be gathering from `src` through the ancestors: nobody may overwrite it (the next
        // observation's dst) before all have finished
        if (!ser_sync<false>(sy, round, xpar, cssm_u128_zero(), cssm_u128_zero(), 0ull, nullptr)) { if (threadIdx.x == 0) atomicOr(&sc->err, 16u); return; }
      } else {
        ser_wait_vm();            // the next observation reads this block's own rows back
      }
      via_anc = false;
      if (stamp) for (int q = 1; q < CSSM_SER_TS_PER_STEP; ++q) ts[(size_t)s * CSSM_SER_TS_PER_STEP + q] = wall_clock64();
      continue;
    }
    cssm_u128 S = acc.S, S2 = acc.S2;
    double tmax = acc.tmax;
    bool bad = acc.bad;
    series_block_reduce(S, S2, tmax, bad);
    if (bad && threadIdx.x == 0) atomicOr(&sc->err, 1u);
    if (stamp) ts[(size_t)s * CSSM_SER_TS_PER_STEP + 1] = wall_clock64();
    // ---- exchange: prefix of the blocks before this one, totals, max
    SerSums X;
    if (!ser_sync<true>(sy, round, xpar, S, S2, cssm_order_key(tmax), &X)) { if (threadIdx.x == 0) atomicOr(&sc->err, 16u); return; }
    const double gmax = cssm_order_unkey(X.key);
    const double cref = rec->ref;
    const double level = cssm_ref_choose(cref, gmax);
    if (!(level == cref)) {
      // the max ruled the observation's reference level out (an outlying observation): the sums again, relative to the max
      // -- every block takes this branch or none does (the decision is a function of the exchanged max alone)
      S = cssm_u128_zero(); S2 = cssm_u128_zero();
      for (uint32_t i = threadIdx.x; i < cnt; i += CSSM_BLOCK) {
        const double w1 = cssm_exp(s_lw[i] - level);
        S = cssm_u128_add(S, cssm_fix_from_double(w1));
        S2 = cssm_u128_add(S2, cssm_fix_from_double(w1 * w1));
      }
      double dummy = tmax; bool b2 = false;
      series_block_reduce(S, S2, dummy, b2);
      if (!ser_sync<true>(sy, round, xpar, S, S2, X.key, &X)) { if (threadIdx.x == 0) atomicOr(&sc->err, 16u); return; }
    }
    if (stamp) ts[(size_t)s * CSSM_SER_TS_PER_STEP + 2] = wall_clock64();
    const bool usable = !cssm_u128_is_zero(X.tot) && (gmax > -cssm_inf()) && (gmax < cssm_inf());
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // :127-128
      sc->gmax = gmax; sc->ref = level; sc->S_off = cssm_u128_zero(); sc->S_local = X.tot; sc->S2_local = X.tot2;
      sc->S_tot = X.tot; sc->S2_tot = X.tot2;
      finish_step(sc, n_arg);                    // (raises err bit 1 (2) itself when the weights are unusable)
      if (ll_t) { ll_t[s] = sc->ll; ess_t[s] = sc->ess; }
    }
    // ---- phase O
    if (usable) {
      series_offspring(s_lw, range_lo, cnt, level, X.pre, X.tot, rec->u, n_arg, anc, force_exact, reinterpret_cast<uint32_t*>(s_buf));
    } else {                                     // all weights zero / max not finite: keep every particle (the host reports the error)
      for (uint32_t i = threadIdx.x; i < cnt; i += CSSM_BLOCK)
        __hip_atomic_store(anc + range_lo + i, range_lo + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp) ts[(size_t)s * CSSM_SER_TS_PER_STEP + 3] = wall_clock64();
    // ---- the ancestors of every slot are in memory before anyone gathers through them
    if (!ser_sync<false>(sy, round, xpar, cssm_u128_zero(), cssm_u128_zero(), 0ull, nullptr)) { if (threadIdx.x == 0) atomicOr(&sc->err, 16u); return; }
    via_anc = true;
    if (stamp) ts[(size_t)s * CSSM_SER_TS_PER_STEP + 4] = wall_clock64();
  }
}

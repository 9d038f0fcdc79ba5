// cssm_propagate.hip.h -- k_propagate, the fused gather + propagate + weight kernel.  Included by
// cssm_prop.hip, which is compiled once per latent dimension D (the 112 instantiations built in parallel).
#pragma once

#if !defined(__HIPCC_RTC__)   /* (hipRTC -- csrc/cssm_rtc.cpp compiles this header at run time for a handle's model structure -- has no system headers) */
#include <cstddef>
#endif

#include "cssm_device.hip.h"

// Particles per thread in k_propagate: all gathers of a thread are issued before its ALU work, and rows are stored as
// 16-byte vectors.  Two per thread for every d <= 8: a wave's store instruction then covers 1 KiB contiguously, which is
// what the write-through (sc1) stores need -- with four per thread (round 1's layout for d <= 2: two 16-byte stores per
// lane, 32 bytes apart) they lost the L2's write combining and the plain stores that layout fell back to left the
// launch's dirty lines to its end: k_propagate<1> 115 -> 93 us at N = 2^24 (0.43 -> 0.54 of the HBM peak), 13.8 -> 11.8 us
// at 2^20.
#ifndef CSSM_PROP_IT_LO
#define CSSM_PROP_IT_LO 2
#endif
// 1 (default): the single GPU's tile-after-tile fused-sums launch gives every wave a contiguous quarter of its block's range (propagate_block);
// 0 (A/B builds): the block-wide tiles of rounds 1-5 -- the host then never asks for the waves' sums and k_offspring_self resamples
#ifndef CSSM_PROP_WR
#define CSSM_PROP_WR 1
#endif
template <int D> struct PropItems { static constexpr int value = (D <= 2) ? CSSM_PROP_IT_LO : (D <= 8 ? CSSM_PROP_IT_MID : 1); };


// stepFilter lines :118 and :123-124 fused (LGCP: calcWeight :184-208).  src is read through
// anc[] when anc != nullptr (the previous step's resampling).  A thread owns IT consecutive
// particles; a block owns CSSM_BLOCK*IT consecutive particles per grid-stride iteration.
// min waves per SIMD asked of the register allocator: 4 for small d (the kernel is VALU-bound and needs
// the co-resident waves to cover LDS/table and gather latency), 3 beyond
#ifndef CSSM_PROP_WAVES_LO
#define CSSM_PROP_WAVES_LO 4
#endif
#ifndef CSSM_PROP_WAVES_SUMS
#define CSSM_PROP_WAVES_SUMS 4
#endif
// (the kernels that also form the sums need ~10 more VGPRs; at 4 waves they spill 12 bytes, and a scratch reload in the
// compute phase waits for the prefetch like any other vector-memory operation -- measured all the same: 4 waves with
// that spill 41.2 us/step at N = 2^20 and 375 us at 2^24, 3 waves without it 41.9 and 383)
template <int D, int SUMS = 0> struct PropWaves { static constexpr int value = (D <= 4) ? (SUMS ? CSSM_PROP_WAVES_SUMS : CSSM_PROP_WAVES_LO) : 3; };

// Diagnostic build only (-DCSSM_PROP_STAMPS, tools/archive/propagate_stamps.py): lane 0 of wave 0 of every block of the slim kernels leaves the
// 100 MHz clock at the points marked PSTAMP (8 words per block).
// (Measured with these stamps and dropped: s_setprio by phase -- first tile's normals 3, its arithmetic 2, second tile's normals 1, the
//  rest 0 -- so that the waves of a SIMD, which the arbiter otherwise serves oldest first, progress together: they do (blocks done
//  at 9.3 .. 12.9 us instead of 6.4 .. 12.5), and the kernel is 0.4 us SLOWER: the pipe was busy either way.)
#ifdef CSSM_PROP_STAMPS
static __device__ unsigned long long g_prop_stamps[8192 * 8];
#define PSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prop_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif

// LDS staging area of propagate_range: per wave IT * D regions of 64 lanes x ES bytes (ES = 16: one dwordx4 fetch per lane and
// element, of which the first 8 bytes are the element, while 4 blocks of that fit the CU's 160 KiB; else 8: two dword fetches).
template <int D, int IT> struct PropStage {
  static constexpr int ES = (IT * D <= 8) ? 16 : 8;
  static constexpr int wave_bytes = IT * D * 64 * ES;
  static constexpr int bytes = (CSSM_BLOCK / 64) * wave_bytes;
};

// Thread-local results of propagate_range, reduced over the block by its caller.
struct PropAcc {
  cssm_u128 S, S2;   // fixed-point sums of exp(w - c) (SUMS >= 1) and of exp(w - c)^2 (SUMS == 2) over the thread's particles
  double tmax;       // largest log-weight seen
  bool bad;          // a log-weight was NaN
};

// The body of every k_propagate launch: the block's particles [range_lo, n) of one observation.
// SUMS: what the kernel does with a weighted particle's log-weight w --
//   0  stores w (the multinomial resampler; an observation that is redone; the first event of an LGCP series, whose level is its
//      max: the sums are then a pass of their own, k_tile_sums);
//   1  forms w1 = exp(min(w - c, 2^-20)) relative to the observation's reference level c, adds it to the block's fixed-point
//      sum S and stores W1 IN PLACE OF w (same 8 bytes): k_offspring then needs no exp, only the conversion the sum was
//      formed with.  Sum of squares: k_offspring's business (nothing on the device depends on the ESS).  Single GPU;
//   2  the same, and also S2 = sum w1^2 (the sharded exchange ships both sums in its segment headers).
// `tab`: the contract's log table, staged in LDS by the caller (stage_log_table).
// ONE (small clouds): the range is a single tile, and the tile's normal variates -- which depend on nothing but (seed, particle,
// observation) -- are drawn WHILE the ancestor indices and the gathered rows are on their way: with one wave per SIMD nobody
// else hides those two round trips (~1.7 us of a kernel whose whole body takes ~5).  Same arithmetic, another order.
// (ONE = 1: the range is a single tile; ONE = 2: the same body tile after tile -- a separate instantiation: folding the loop into
// the single-tile kernel cost it 18 VGPRs and 10 % at d = 9)
// WR (wave ranges; round 6): the range [range_lo, n) is ONE WAVE's -- a contiguous quarter of its block's unit, walked in tiles of 64 * IT
// particles, lane l owning the IT particles at 64 IT tile + IT l -- instead of the whole block's (tiles of CSSM_BLOCK * IT, thread t at IT t).
// Every particle's arithmetic depends on its global id alone and the sums are integers: the same bits either way.  What it buys: the wave's
// sum, which the block reduction forms anyway, is the exact sum of a contiguous quarter unit -- k_offspring_wave's waves take their
// prefixes from those sums instead of converting and scanning every weight again (cssm_offspring_wave.hip.h).  A wave's store
// instruction covers 1 KiB contiguously as before; the block streams four ranges instead of one.
template <int D, bool LGCP, int IT, int OBS, int SUMS, int ONE = 0, bool WR = false>
__device__ __forceinline__ void propagate_range(
    const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc,
    double* __restrict__ dst, size_t dst_stride, double* __restrict__ logw, uint64_t gid0,
    uint64_t seed, const StepRec* __restrict__ rec, const ModelK& mk,
    const double* __restrict__ src2, size_t src2_stride, uint32_t n_split, const double* tab,
    const uint32_t range_lo, const uint32_t n, int do_sums_arg,
    double* __restrict__ pick_out, uint32_t pick_slot, unsigned char* s_stage, PropAcc& acc,
    const double* __restrict__ fsub = nullptr, const unsigned long long* pre_jp = nullptr, const cssm_u32x4* pre_blk = nullptr,
    const uint32_t step_now = 0u, const bool wr_on = false) {
  // pre_jp (ONE): the tile's packed ancestor indices, already requested by the caller (before it staged the log table: one
  // dependent round trip less)
  // pre_blk (ONE, pairs): the Philox blocks of the first tile's pairs, drawn by the caller while those first loads travelled
  // (step_now = the observation's index as the HOST passed it: rec->step is itself a first load)
  // fsub (LGCP with a time-dependent f, e.g. a seasonal leaf): the handle's table of f coefficients at the sub-step times
  // tau_s = t + s delta (FilterLgcp.calcWeight evaluates mod.f(a.state, a.time) at every simulated time,
  // model/ParticleFilter.scala:193-205; model/Sde.scala:57-66); this observation's rows start at rec->fsub_off
  // s_stage: PropStage<D, IT>::bytes bytes of LDS (16-byte aligned) owned by the caller
  // SUMS && pick_out != nullptr (`filter`, model/ParticleFilter.scala:157): the thread that gathers slot pick_slot holds the
  // resampled state sampleOne chose after the PREVIOUS observation, before its transition: it records it (a launch of
  // its own per observation would cost more than the whole sums pass at small N)
  // (Totalling the sub-unit sums in the block that finishes last -- the threadfence-reduction idiom -- was measured
  // and rejected: on this multi-XCD part every block's device-scope release fence writes the L2's dirty lines back,
  // which in a kernel that streams hundreds of MB of stores cost 130 us at N = 2^24.  k_scan_tiles does it instead.)
  // src2 != nullptr (sharded filter): ancestor indices >= n_split address the candidates received from
  // other ranks, src2[k * src2_stride + (j - n_split)]
  const uint32_t step = (pre_blk != nullptr) ? step_now : rec->step;
  const int has_obs = rec->has_obs;
  const double dt = rec->dt;
  const bool weighted = LGCP || has_obs;
  // every thread's first particle has an even global id (gid0 even; chunk, tile and IT even): whole pairs per thread
  const bool pair_ok = (gid0 & 1ull) == 0ull;
  // SUMS (compile time: its accumulators would otherwise hold 8 VGPRs in every kernel): the block also forms the sums
  const bool do_sums = SUMS && do_sums_arg && weighted;   // (do_sums_arg is 1 for every SUMS instantiation; LGCP: the level is predicted, contract v8)
  const double cref = rec->ref;
  cssm_u128 accS = cssm_u128_zero(), accS2 = cssm_u128_zero();
  double tmax = -cssm_inf();
  bool bad = false;
  // tile indices are 32-bit (the library admits n <= 2^32 - 2^16 particles per handle): half the VALU work of 64-bit
  static_assert(!WR || ONE == 2, "wave ranges: the tile-after-tile instantiation");
  const bool wr = WR && wr_on;                                              // (uniform: the launch's choice, propagate_block)
  const uint32_t stride = (wr ? 64u : (uint32_t)CSSM_BLOCK) * IT;           // particles per tile (wave ranges: of the wave)
  const uint32_t tl = wr ? (threadIdx.x & 63u) : threadIdx.x;               // the thread's place in its tile
  // Software pipeline over the block's tiles WITHOUT spending registers on it.  With 4 waves per SIMD the two
  // dependent memory round trips of a tile (ancestor index -> gathered state) are exposed: a model with that
  // latency reproduces the 3.3 TB/s the un-pipelined kernel reached for every d, and holding the next tile in VGPRs
  // costs a wave of occupancy (measured: slower).  Instead the NEXT tile's states are fetched by asynchronous
  // global -> LDS loads (global_load_lds_dwordx4, a gfx950 instruction: each lane fetches the 16 bytes at its own
  // address into slot lane * 16 of a 1 KiB LDS region, no VGPR is written) while the current tile is computed, and the
  // ancestor indices of the tile after that are fetched into the IT index registers.  A wave reads back only what it
  // loaded itself, so no block barrier is involved.  Per wave: IT * D regions of 1 KiB.
  // Bytes per lane and element: 16 (one dwordx4 fetch, of which the first 8 bytes are the element) while 4 blocks of
  // that fit the CU's 160 KiB of LDS next to the 6 KiB contract table (v7: d = 9 no longer does), else 8 (two dword fetches:
  // low and high word).
  constexpr bool STAGE = true;
  constexpr int ES = (IT * D <= 8) ? 16 : 8;
  constexpr int WAVE_STAGE = IT * D * 64 * ES;
  unsigned char* const wstage = s_stage + (size_t)(threadIdx.x >> 6) * WAVE_STAGE;
  const uint32_t wstage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)wstage);
  const uint32_t n_last = n - 1u;
  // indices of this thread's IT particles of the tile at `base`, clamped into range (stores are predicated)
  // One vector load for every thread (anc has `stride` >= n entries rounded up to a tile, so the load of a partial or
  // empty thread stays inside the buffer).  The indices stay PACKED two to a 64-bit register exactly as they were
  // loaded, and nothing touches them until they are consumed: any operation on them next to the load -- even the
  // register copy that unpacking a vector load can need -- makes the compiler wait for the load right there.
  constexpr int NJ = (IT + 1) / 2;
  auto load_idx = [&](uint32_t base, unsigned long long (&jp)[NJ]) {
    const uint32_t i0 = base + tl * IT;
    if (anc) {
      if (IT == 4) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(anc + i0);
        jp[0] = a.x; jp[NJ - 1] = a.y;
      } else if (IT == 2) {
        jp[0] = *reinterpret_cast<const unsigned long long*>(anc + i0);
      } else {
        jp[0] = anc[i0];
      }
    } else {
#pragma unroll
      for (int q = 0; q < NJ; ++q) jp[q] = (unsigned long long)(uint32_t)(i0 + 2 * q) | ((unsigned long long)(uint32_t)(i0 + 2 * q + 1) << 32);
    }
  };
  // unpacked where they are consumed; what a partial thread read beyond n is replaced by a valid index
  auto unpack_idx = [&](uint32_t base, const unsigned long long (&jp)[NJ], uint32_t (&j)[IT]) {
    const uint32_t i0 = base + tl * IT;
#pragma unroll
    for (int r = 0; r < IT; ++r) {
      const uint32_t v = (uint32_t)(jp[r / 2] >> (32 * (r & 1)));
      j[r] = (i0 + r < n) ? v : n_last;
    }
  };
  // src2_stride == 0: the candidates are rows of (D + 1) doubles (state, end slot) exactly as they were received
  // (single-collective exchange: the receive buffer is read in place); otherwise struct-of-arrays with that stride
  auto ptr_of = [&](uint32_t j, int k) -> const double* {
    if (src2 && j >= n_split)
      return (src2_stride == 0) ? src2 + (size_t)(j - n_split) * (size_t)(D + 1) + k : src2 + (size_t)k * src2_stride + (j - n_split);
    return src + (size_t)k * src_stride + j;
  };
  auto gather = [&](const uint32_t (&j)[IT], double (&x)[IT][D]) {
#pragma unroll
    for (int r = 0; r < IT; ++r)
#pragma unroll
      for (int k = 0; k < D; ++k) x[r][k] = (src2 && j[r] >= n_split) ? ld_sys_f64(ptr_of(j[r], k)) : *ptr_of(j[r], k);
  };
  // (every gather source is allocated with 16 spare bytes: the 16-byte fetch of the last element of a buffer stays inside it)
  auto stage_issue = [&](const uint32_t (&j)[IT]) {
    // (sharded filter) rows received from the neighbouring ranks are gathered by the slots next to the rank's two ends only -- ancestors
    // are monotone in the slot --: a wave none of whose indices reaches them (wave-uniform test) addresses the local cloud alone, with
    // the row base in scalar registers, instead of forming both addresses and a select per element (~25 VALU instructions per particle
    // at d = 3: most of what k_propagate_shard cost beyond k_propagate_self)
    bool local_only = (src2 == nullptr);
    if (src2 != nullptr) {
      uint32_t jm = j[0];
#pragma unroll
      for (int r = 1; r < IT; ++r) jm = (j[r] > jm) ? j[r] : jm;
      local_only = !__any(jm >= n_split);
    }
#pragma unroll
    for (int r = 0; r < IT; ++r)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        // (src2 == nullptr is uniform: without candidates from other ranks the row base stays in scalar registers)
        const double* g = local_only ? src + (size_t)k * src_stride + j[r] : ptr_of(j[r], k);
        const uint32_t slot = wstage_lds + (uint32_t)((r * D + k) * 64 * ES);
        if (!local_only && j[r] >= n_split) {
          // a row another rank wrote into this rank's receive window (or an all-to-all delivered): read with a SYSTEM-scope load, like
          // everything else that reads a window (ld_sys), and put into the lane's staging slot by hand -- what this GPU's caches may
          // still hold of the window's previous use never answers it.  The few lanes next to the rank's ends take this path.
          const double v = ld_sys_f64(g);
          unsigned char* sl = wstage + (size_t)((r * D + k) * 64 * ES);
          const uint32_t lane = threadIdx.x & 63;
          if (ES == 16) {
            *reinterpret_cast<double*>(sl + lane * 16) = v;
          } else {
            *reinterpret_cast<uint32_t*>(sl + lane * 4) = (uint32_t)cssm_d2u(v);
            *reinterpret_cast<uint32_t*>(sl + 256 + lane * 4) = (uint32_t)(cssm_d2u(v) >> 32);
          }
        } else
        if (ES == 16) {
          lds_dma16(g, slot);
        } else {
          lds_dma4(g, slot);
          lds_dma4(reinterpret_cast<const unsigned char*>(g) + 4, slot + 256u);
        }
      }
  };
  auto stage_read = [&](double (&x)[IT][D]) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < IT; ++r)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const unsigned char* slot = wstage + (r * D + k) * 64 * ES;
        if (ES == 16) {
          x[r][k] = *reinterpret_cast<const double*>(slot + lane * 16);
        } else {
          const uint32_t lo = *reinterpret_cast<const uint32_t*>(slot + lane * 4);
          const uint32_t hi = *reinterpret_cast<const uint32_t*>(slot + 256 + lane * 4);
          x[r][k] = cssm_u2d((uint64_t)lo | ((uint64_t)hi << 32));
        }
      }
  };
  uint32_t base = range_lo;
  unsigned long long jp[NJ];
  uint32_t jn[IT];
  double x[IT][D];
  static_assert(!ONE || (IT <= 2 && !LGCP), "ONE: one pair (d <= 8) or one particle (d >= 9) per thread, ordinary step");
  double zz[ONE ? IT * D : 1];                                    // ONE: the thread's IT * D normals (normal q -> particle q / D, component q % D)
  if (base < n) {
    if (NJ == 1 && pre_jp != nullptr) jp[0] = *pre_jp; else load_idx(base, jp);   // (requested by the caller with its other first loads)
    if (ONE && IT == 2 && pre_jp == nullptr) {                    // (while the indices travel)
      normals_pair_half<D, 0>(seed, gid0 + base + tl * IT, step, tab, zz);
      // the indices are consumed BEHIND these normals: an empty asm that takes both pins the order (the compiler otherwise
      // hoists the address arithmetic, and with it the wait for the load, above the Philox rounds)
#pragma unroll
      for (int q = 0; q < PairHalf<D>::n0; ++q) asm volatile("" : "+v"(zz[q % (ONE ? IT * D : 1)]), "+v"(jp[0]));
      if (PairHalf<D>::n0 == 0) asm volatile("" : "+v"(jp[0]));
    }
    unpack_idx(base, jp, jn);
    if (STAGE) {
      stage_issue(jn);                                            // tile 0 (needs its indices: the one exposed latency)
      if (!ONE && base + stride < n) load_idx(base + stride, jp);         // indices of tile 1
      if (ONE && IT == 2) {                                       // (while the rows travel)
        if (pre_blk != nullptr) {
          normals_pair_from_blocks<D>(pre_blk, tab, zz);
        } else {
          if (pre_jp != nullptr) normals_pair_half<D, 0>(seed, gid0 + base + tl * IT, step, tab, zz);
          normals_pair_half<D, 1>(seed, gid0 + base + tl * IT, step, tab, zz);
        }
#pragma unroll
        for (int q = (pre_jp != nullptr) ? 0 : PairHalf<D>::n0; q < 2 * D; ++q) asm volatile("" : "+v"(zz[q % (ONE ? IT * D : 1)]));
      } else if (ONE) {                                           // one particle per thread: its D normals
        double z1[D];
        draw_normals<D>(seed, gid0 + base + tl, step, CSSM_STREAM_STEP, tab, z1);
#pragma unroll
        for (int q = 0; q < D; ++q) { zz[q % (ONE ? IT * D : 1)] = z1[q]; asm volatile("" : "+v"(zz[q % (ONE ? IT * D : 1)])); }
      }
      if (ONE) PSTAMP(2);                                         // (normals drawn)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stage_read(x);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the region is free again
      if (ONE) PSTAMP(3);                                         // (rows landed)
      if (!ONE && base + stride < n) {
        unpack_idx(base + stride, jp, jn);
        stage_issue(jn);                                          // tile 1 lands while tile 0 is computed
        if (base + 2 * stride < n) load_idx(base + 2 * stride, jp);
      }
    } else {
      gather(jn, x);
    }
  }
  for (; base < n; base += stride) {
    const uint32_t i0 = base + tl * IT;
    const bool full = (i0 + IT <= n);
    // (tile after tile: the NEXT tile's ancestor indices travel while this one is computed -- one of the two dependent round trips
    //  at the tile boundary; two registers)
    if (ONE == 2 && base + stride < n) load_idx(base + stride, jp);
    if (SUMS && pick_out != nullptr) {   // (only the SUMS kernels carry this: it costs the lean kernel 2 % for nothing)
#pragma unroll
      for (int r = 0; r < IT; ++r)
        if (i0 + r == pick_slot) {
#pragma unroll
          for (int k = 0; k < D; ++k) pick_out[k] = x[r][k];
        }
    }
    double lw[IT];
    // weight of particle r once its state is propagated: NaN check, running max, optional fused sums
    auto account = [&](int r) {
      if (weighted && i0 + r < n) {
        if (lw[r] != lw[r]) { bad = true; lw[r] = -cssm_inf(); }
        tmax = (lw[r] > tmax) ? lw[r] : tmax;
        if (SUMS && do_sums) {
          // beyond c + CSSM_REF_BELOW the step is redone with the max anyway: keep the conversion in range
          // (a <= 2^-20 and never NaN: the cheaper forms of exp and of the fixed-point conversion return the same values)
          const double a = cssm_min_c(lw[r] - cref, CSSM_REF_BELOW);
          const double w1 = cssm_exp_le0(a);
          accS = cssm_u128_add(accS, cssm_fix_from_unit(w1));
          if (SUMS == 2) accS2 = cssm_u128_add(accS2, cssm_fix_from_unit(w1 * w1));
          lw[r] = w1;                            // what is stored: the weight, not its logarithm
        }
      }
    };
    if (LGCP) {
      // (measured and dropped: the thread's particles walking the sub-steps side by side -- loop over s outside -- for
      //  instruction-level parallelism between their dependent chains: 459 vs 463 us at N = 2^24, five more registers)
#pragma unroll
      for (int r = 0; r < IT; ++r) {
        double z[D];
        const uint64_t gid = gid0 + i0 + r;
        const int nsub = rec->n_sub;
        if (nsub == 0) {                     // dt == 0: (x, f, f), model/ParticleFilter.scala:212-213
          double g = gamma_of<D>(mk, rec, x[r]);
          lw[r] = g - g;
        } else {
          double haz = 0.0, carry = 0.0;
          uint32_t held_a = 0u, held_b = 0u;     // the second half of the Philox block in use (contract v6: a block = two pairs)
          const double* fsub_obs = (fsub != nullptr) ? fsub + rec->fsub_off : nullptr;
          for (int s = 0; s < nsub; ++s) {   // simInitStream(...).take(n), :193-194
            // normal number q = s*D + k: even q opens Box-Muller pair q>>1 (second element kept for q+1); pair P is half P & 1
            // of Philox block P >> 1: q = 0 mod 4 draws a block, q = 2 mod 4 uses the half it kept
#pragma unroll
            for (int k = 0; k < D; ++k) {
              const uint32_t q = (uint32_t)s * D + k;
              if ((q & 1u) == 0u) {
                double z0, z1;
                if ((q & 2u) == 0u) {
                  const cssm_u32x4 blk = cssm_philox_draw(seed, gid, step, CSSM_STREAM_STEP, q >> 2);
                  held_a = blk.v[2]; held_b = blk.v[3];
                  cssm_normal_pair64(blk.v[0], blk.v[1], tab, &z0, &z1);
                } else {
                  cssm_normal_pair64(held_a, held_b, tab, &z0, &z1);
                }
                z[k] = z0; carry = z1;
              } else {
                z[k] = carry;
              }
            }
            transition<D>(mk, rec, dt, x[r], z);
            const double* fc = (fsub_obs != nullptr) ? fsub_obs + (size_t)s * D : rec->fco;   // f at tau_s (uniform)
            haz = haz + cssm_exp(gamma_coef<D>(mk, fc, x[r])) * dt;   // :203-205
          }
          lw[r] = gamma_of<D>(mk, rec, x[r]) - haz;                // :200,:217
        }
        account(r);
      }
    } else if (ONE) {
      // (gid0 is even for every caller of this instantiation: whole pairs)
#pragma unroll
      for (int q = 0; q < IT * D; ++q) transition_one<D>(mk, rec, dt, q % D, x[(q / D) % IT][q % D], zz[q % (ONE ? IT * D : 1)]);
#pragma unroll
      for (int r = 0; r < IT; ++r) {
        lw[r] = has_obs ? logdens<OBS>(mk, rec, gamma_of<D>(mk, rec, x[r]), tab) : 0.0;
        account(r);
      }
    } else if (IT % 2 == 0 && pair_ok) {
      // the thread's particles are whole pairs (2m, 2m+1): D Philox blocks + Box-Muller pairs per two particles
#pragma unroll
      for (int r = 0; r + 1 < IT; r += 2) {
        propagate_pair<D>(mk, rec, dt, seed, gid0 + i0 + r, step, tab, x[r], x[(r + 1) % IT]);
        lw[r] = has_obs ? logdens<OBS>(mk, rec, gamma_of<D>(mk, rec, x[r]), tab) : 0.0;
        account(r);
        lw[(r + 1) % IT] = has_obs ? logdens<OBS>(mk, rec, gamma_of<D>(mk, rec, x[(r + 1) % IT]), tab) : 0.0;
        account((r + 1) % IT);
      }
    } else {
#pragma unroll
      for (int r = 0; r < IT; ++r) {
        propagate_one<D>(mk, rec, dt, seed, gid0 + i0 + r, step, tab, x[r]);
        lw[r] = has_obs ? logdens<OBS>(mk, rec, gamma_of<D>(mk, rec, x[r]), tab) : 0.0;
        account(r);
      }
    }
    // everything older is complete by now without having been waited for: the next tile's states (issued one tile of
    // compute ago), the indices of the tile after it, and the previous tile's stores
    if (STAGE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the index registers are complete as well: tell the compiler here, or it waits for them (and with them for the
      // stores below) where the next tile's loads are issued
#pragma unroll
      for (int q = 0; q < NJ; ++q) asm volatile("" : "+v"(jp[q]));
    }
    if (full && IT == 4) {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        double* p = dst + (size_t)k * dst_stride + i0;
        *reinterpret_cast<double2*>(p) = make_double2(x[0][k], x[1 % IT][k]);
        *reinterpret_cast<double2*>(p + 2) = make_double2(x[2 % IT][k], x[3 % IT][k]);
      }
      if (weighted) {
        *reinterpret_cast<double2*>(logw + i0) = make_double2(lw[0], lw[1 % IT]);
        *reinterpret_cast<double2*>(logw + i0 + 2) = make_double2(lw[2 % IT], lw[3 % IT]);
      }
    } else if (full && IT == 2) {
#pragma unroll
      for (int k = 0; k < D; ++k)
        bulk_store2(dst + (size_t)k * dst_stride + i0, x[0][k], x[1 % IT][k]);
      if (weighted && logw) bulk_store2(logw + i0, lw[0], lw[1 % IT]);
    } else {
#pragma unroll
      for (int r = 0; r < IT; ++r) {
        if (i0 + r < n) {
#pragma unroll
          for (int k = 0; k < D; ++k) dst[(size_t)k * dst_stride + i0 + r] = x[r][k];
          if (weighted && logw) logw[i0 + r] = lw[r];
        }
      }
    }
    if (ONE && base == range_lo) PSTAMP(4);                      // (first tile computed, stores issued)
    if (ONE == 1) break;                                          // (the range is this one tile)
    if (ONE == 2) {   // further tiles of the range in the same way: no software pipeline, co-resident waves cover the round trips
      const uint32_t nb = base + stride;
      if (nb < n) {
        unpack_idx(nb, jp, jn);
        stage_issue(jn);
        if (IT == 2) {
          normals_pair_half<D, 0>(seed, gid0 + nb + tl * IT, step, tab, zz);
          normals_pair_half<D, 1>(seed, gid0 + nb + tl * IT, step, tab, zz);
        } else {
          double z1[D];
          draw_normals<D>(seed, gid0 + nb + tl, step, CSSM_STREAM_STEP, tab, z1);
#pragma unroll
          for (int q = 0; q < D; ++q) zz[q % (ONE ? IT * D : 1)] = z1[q];
        }
#pragma unroll
        for (int q = 0; q < IT * D; ++q) asm volatile("" : "+v"(zz[q % (ONE ? IT * D : 1)]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_read(x);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (base == range_lo) PSTAMP(5);                         // (second tile's rows landed)
      }
      continue;
    }
    // advance the pipeline
    if (STAGE) {
      if (base + stride < n) {
        stage_read(x);                                            // tile i + 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (base + 2 * stride < n) {
          unpack_idx(base + 2 * stride, jp, jn);
          stage_issue(jn);                                        // tile i + 2
          if (base + 3 * stride < n) load_idx(base + 3 * stride, jp);
        }
      }
    } else if (base + stride < n) {
      load_idx(base + stride, jp);
      unpack_idx(base + stride, jp, jn);
      gather(jn, x);
    }
  }
  acc.S = accS; acc.S2 = accS2; acc.tmax = tmax; acc.bad = bad;
}

//
// A block owns the CONTIGUOUS range [blockIdx.x * chunk, +chunk) (chunk a multiple of CSSM_BLOCK*IT, chosen on the
// host so that a whole number of blocks makes one scan unit of k_offspring).  With do_sums the block also forms
// S = sum exp(w - c), S2 = sum exp(w - c)^2 in fixed point for its range, c = rec->ref being known before any
// weight is (include/cssm_numerics.h, "reference level"): the log-sum-exp of :125-127 then needs no pass of its own.
template <int D, bool LGCP, int IT, int OBS, int SUMS, uint32_t MKW = 0u, uint32_t MKW1 = 0u, uint32_t MKW2 = 0u>
__global__ __launch_bounds__(CSSM_BLOCK, (PropWaves<D, SUMS>::value)) void k_propagate(
    const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc,
    double* __restrict__ dst, size_t dst_stride, double* __restrict__ logw, uint64_t n_arg, uint64_t gid0,
    uint64_t seed, const StepRec* __restrict__ rec, ModelK mk, Scalars* __restrict__ sc, int slot_set_arg,
    const double* __restrict__ src2, size_t src2_stride, uint32_t n_split, const double* __restrict__ logtab,
    uint64_t chunk, int do_sums_arg, cssm_u128* __restrict__ subS, cssm_u128* __restrict__ subS2,
    double* __restrict__ pick_out, uint32_t pick_slot, const double* __restrict__ fsub) {
  __shared__ double s_max[CSSM_BLOCK / 64];
  if (MKW != 0u) { mk.comp[0] = MKW; mk.comp[1] = MKW1; mk.comp[2] = MKW2; }   // (a known structure at compile time: k_propagate_self)
  // slot_set_arg: as in propagate_block -- bits 0-7 the set of max slots, bit 8 + bits 9-10: one block per unit, group sums wanted, their set
  const int slot_set = slot_set_arg & 0xff;
  const bool grp_on = (slot_set_arg & 0x100) != 0;
  const int grp_set = (slot_set_arg >> 9) & 3;
  const int grp_shift = 5 + ((slot_set_arg >> 11) & 3);
  // (sharded series) the exchange of an earlier observation did not fit: the series is on hold and nothing may change
  // until the host resumes it (cssm_pf_shard_resume)
  // bit 3: a sharded series is on hold (capacity miss); bit 6: a single-GPU batch series waits for the redo of an outlying
  // observation; bit 2: a series enqueued ahead is void (its level was ruled out: the host repeats it) -- nothing to do
  // ONE round of first loads: the hold word, the contract table's three entries of this thread and the first tile's ancestor indices are
  // requested together (they were three dependent round trips -- ~2.5 us per block, which a launch whose blocks all run in one round,
  // an LGCP shard of 2^21 particles, pays in full)
  const uint32_t held = sc->err;
  const uint32_t range_lo = blockIdx.x * (uint32_t)chunk;
  uint32_t n;                                                 // this block's range ends at n
  { const uint64_t range_hi = (uint64_t)range_lo + chunk; n = (uint32_t)((range_hi < n_arg) ? range_hi : n_arg); }
  constexpr bool EARLY_IDX = (IT == 2);                        // (one packed index register: propagate_range's pre_jp)
  unsigned long long jp_early = 0ull;
  if (EARLY_IDX && range_lo < n) {
    const uint32_t i0 = range_lo + threadIdx.x * IT;           // (anc holds a whole number of tiles: a partial thread's pair stays inside it)
    jp_early = anc ? *reinterpret_cast<const unsigned long long*>(anc + i0) : ((unsigned long long)i0 | ((unsigned long long)(i0 + 1u) << 32));
  }
  static_assert(CSSM_BLOCK == 256, "three table entries per thread");
  const double tv = logtab[threadIdx.x], tv1 = logtab[256 + threadIdx.x], tv2 = logtab[512 + threadIdx.x];
  const double* tab = stage_log_table_finish(tv, tv1, tv2);
  if (held & (4u | 8u | 16u | 64u)) return;
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[PropStage<D, IT>::bytes];
  PropAcc acc;
  propagate_range<D, LGCP, IT, OBS, SUMS>(src, src_stride, anc, dst, dst_stride, logw, gid0, seed, rec, mk, src2, src2_stride,
                                          n_split, tab, range_lo, n, do_sums_arg, pick_out, pick_slot, s_stage, acc, LGCP ? fsub : nullptr,
                                          (EARLY_IDX && range_lo < n) ? &jp_early : nullptr);
  const bool weighted = LGCP || rec->has_obs;
  const bool do_sums = SUMS && do_sums_arg && weighted;
  cssm_u128 accS = acc.S, accS2 = acc.S2;
  double tmax = acc.tmax;
  const bool bad = acc.bad;
  if (!weighted) return;
  tmax = wave_max(tmax);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = tmax;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&sc->err, 1u);
  if (SUMS && do_sums) {
    __shared__ cssm_u128 s_sa[CSSM_BLOCK / 64], s_sb[CSSM_BLOCK / 64];
    accS = wave_sum_u128(accS);
    if (SUMS == 2) accS2 = wave_sum_u128(accS2);
    if ((threadIdx.x & 63) == 0) { s_sa[threadIdx.x >> 6] = accS; if (SUMS == 2) s_sb[threadIdx.x >> 6] = accS2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      cssm_u128 ta = s_sa[0], tb = (SUMS == 2) ? s_sb[0] : cssm_u128_zero();
#pragma unroll
      for (int w = 1; w < CSSM_BLOCK / 64; ++w) { ta = cssm_u128_add(ta, s_sa[w]); if (SUMS == 2) tb = cssm_u128_add(tb, s_sb[w]); }
      subS[blockIdx.x] = ta;
      if (SUMS == 2) subS2[blockIdx.x] = tb;
      if (grp_on) group_sums_add(sc, grp_set, blockIdx.x >> grp_shift, ta, tb, SUMS == 2);   // (uniform)
    }
  } else {
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double m = s_max[0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) m = (s_max[w] > m) ? s_max[w] : m;
    // one integer atomicMax per block, spread over CSSM_MAXSLOTS cache lines: same-address atomics
    // serialise at ~12 ns each, which at thousands of blocks would cost more than the kernel
    atomicMax(&sc->maxslot[((size_t)slot_set * CSSM_MAXSLOTS + blockIdx.x % CSSM_MAXSLOTS) * CSSM_SLOT_STRIDE],
              (unsigned long long)cssm_order_key(m));
  }
}

// The single-GPU filter's launch of the lean kernel (no fused sums, not LGCP): only the arguments that path uses, and what
// it knows at compile time -- the first particle's global id is 0 (every thread owns whole pairs: the unpaired variant of
// the transition code is not even compiled), no second gather source, no sub-step table, no pick.  The generic kernel's
// ~24 arguments overflow the scalar registers into vector-register lanes (88 v_readlane per tile).
// ONE: the block's range is one tile (small clouds, half a tile per block): see propagate_range
// (the body, shared with the sharded filter's slim launch k_propagate_shard: gid0 = the rank's first global particle, src2 /
//  n_split = the rows received from the neighbouring ranks; both compile-time constants -- 0, nullptr -- in k_propagate_self)
template <int D, int IT, int OBS, int SUMS, int ONE>
__device__ __forceinline__ void propagate_block(
    const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc, double* __restrict__ dst, size_t dst_stride,
    double* __restrict__ logw, uint64_t n_arg, uint64_t seed, const StepRec* __restrict__ rec, const ModelK& mk, Scalars* __restrict__ sc,
    int slot_set_arg, const double* __restrict__ logtab, uint64_t chunk,
    cssm_u128* __restrict__ subS, cssm_u128* __restrict__ subS2, double* __restrict__ pick_out, uint32_t pick_slot,
    const uint64_t gid0, const double* __restrict__ src2, const uint32_t n_split, const uint32_t step_now) {
  // slot_set_arg: bits 0-7 the set of max slots; bit 8: this launch has one block per unit -- the blocks also accumulate the sums of
  // groups of units (Scalars::grp; a shard's launch, SUMS == 2: Scalars::grp2 as well) in set bits 9-10 (the single GPU passes the set
  // of its max slots there; a shard's max slots stay in set 0 while its group sums rotate)
  const int slot_set = slot_set_arg & 0xff;
  const bool grp_on = (slot_set_arg & 0x100) != 0;
  const int grp_set = (slot_set_arg >> 9) & 3;
  const int grp_shift = 5 + ((slot_set_arg >> 11) & 3);    // log2(blocks per group): 32 units x (1, 2 or 4 blocks per unit: bits 11-12)
  // SUMS: the block also forms its fixed-point sums of exp(w - c) (subS / subS2, one entry per block) and, for `filter`,
  // records the state sampleOne picked after the previous observation (pick_out / pick_slot; see k_propagate)
  __shared__ double s_max[CSSM_BLOCK / 64];
  // (ONE, pairs: the Philox key and the observation's index are wanted BEFORE the first loads from memory are waited for -- scalar
  //  loads return out of order, one wait covers all of them -- so they are fetched with the kernel's first arguments)
  uint32_t key_lo = (uint32_t)seed, key_hi = (uint32_t)(seed >> 32), step_k = step_now;
  if (ONE != 0 && IT == 2) asm volatile("" : "+s"(key_lo), "+s"(key_hi), "+s"(step_k));
  PSTAMP(0);
  const uint32_t held = sc->err;          // (tested behind the table staging: its load then overlaps the table's)
  // WR (the single GPU's tile-after-tile launch with the fused sums): every WAVE owns a contiguous quarter of the block's range (see
  // propagate_range); bit 13 of the set argument: the waves' sums are wanted (subS2, unused by SUMS == 1, holds them: four per block)
  // The mapping is the LAUNCH's choice (the same bit): measured against the block-wide tiles it costs this kernel 2.5-3 % at every size
  // (same-box A/B, round 6) and buys k_offspring 11-13 % from 2^21 particles on, nothing at 2^20 -- the host asks for it where it pays.
  constexpr bool WR = CSSM_PROP_WR != 0 && (ONE == 2 && SUMS == 1);
  const bool wsum_on = WR && (slot_set_arg & 0x2000) != 0;
  uint32_t range_lo = blockIdx.x * (uint32_t)chunk;
  uint32_t n;
  { const uint64_t range_hi = (uint64_t)range_lo + chunk; n = (uint32_t)((range_hi < n_arg) ? range_hi : n_arg); }
  if (wsum_on) {
    const uint32_t qw = (uint32_t)(chunk >> 2);             // (the chunk is a whole number of tiles of 1024: a quarter is whole tiles of 64 IT)
    const uint32_t wlo = range_lo + (threadIdx.x >> 6) * qw;
    const uint32_t whi = ((uint64_t)wlo + qw < (uint64_t)n) ? wlo + qw : n;
    // (wave-uniform values the compiler cannot know to be uniform: into scalar registers, or the tile loop's bounds, bases and
    //  addresses turn into vector arithmetic -- the first version of this mapping cost the kernel 4.5 % at N = 2^20 that way)
    range_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)((wlo < n) ? wlo : n));   // (a wave beyond the cloud's end: an empty range)
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)whi);
  }
  const uint32_t tl = wsum_on ? (threadIdx.x & 63u) : threadIdx.x;
  // ONE: the tile's ancestor indices are requested together with the log table -- they depend on nothing but the thread's
  // position (anc holds a whole number of tiles: a partial thread's pair stays inside it)
  unsigned long long jp_early = 0ull;
  if (ONE) {
    const uint32_t i0 = range_lo + tl * IT;
    // (WR: a wave whose range is empty may stand beyond the index buffer's last tile: it requests nothing)
    const bool in_buf = !wsum_on || range_lo < n;
    if (IT == 2) jp_early = (anc && in_buf) ? *reinterpret_cast<const unsigned long long*>(anc + i0) : ((unsigned long long)i0 | ((unsigned long long)(i0 + 1u) << 32));
    else jp_early = (anc && in_buf) ? (unsigned long long)anc[i0] : (unsigned long long)i0;
  }
  // ... and so are the thread's entry of the log table and the lines of the observation's record the kernel will read: behind
  // the table's barrier the record's lines are hits in the scalar cache instead of a first touch on the critical path.  (The
  // comparison cannot be true and cannot be folded: it keeps the loads alive and in place without an asm statement, which
  // would cost the record its scalar loads altogether.)
  const double* tab;
  constexpr bool EARLY = ONE != 0 && IT == 2;
  cssm_u32x4 blk_early[EARLY ? PairHalf<D>::nblk : 1];
  if (ONE) {
    static_assert(CSSM_BLOCK == 256, "one table entry per thread");
    double tv = logtab[threadIdx.x], tv1 = logtab[256 + threadIdx.x], tv2 = logtab[512 + threadIdx.x];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(rec);
    // (one word of every 64-byte line that holds the scalars, the first D rows of coef[] or the first D entries of fco[])
    constexpr int NLINES = (int)((sizeof(StepRec) + 63) / 64);
    auto line_wanted = [](int i) {
      const int o = i * 16;
      return i == 0 || (o * 4 + 4 <= (int)sizeof(StepRec) &&
                        (o * 4 < (int)(offsetof(StepRec, coef) + D * sizeof(double[4]) + 64) ||
                         (o * 4 + 64 > (int)offsetof(StepRec, fco) && o * 4 < (int)(offsetof(StepRec, fco) + D * sizeof(double) + 64))));
    };
    uint32_t pw[NLINES];
#pragma unroll
    for (int i = 0; i < NLINES; ++i) pw[i] = line_wanted(i) ? w[i * 16] : 0u;
    if (EARLY) __builtin_amdgcn_sched_barrier(0);   // (the loads above are issued HERE: the scheduler otherwise sinks them below the rounds)
    // ... and while all of that travels, the Philox blocks of the tile's pairs: they depend on nothing but (seed, pair, observation)
    // -- the observation's index as a kernel argument, rec->step being one of the loads.  (The empty asm statements take a block
    // and a loaded value: the rounds cannot sink below them, the waits for the loads cannot rise above them -- scalar loads return
    // out of order, so a wait for any of them is a wait for all.)
    if (EARLY) {
      const uint64_t stream = cssm_pair_stream(gid0 + range_lo + tl * IT);
#pragma unroll
      for (int B = 0; B < PairHalf<D>::nblk; ++B) {
        blk_early[B] = cssm_philox_draw((uint64_t)key_lo | ((uint64_t)key_hi << 32), stream, step_k, CSSM_STREAM_STEP, (uint32_t)B);
        asm volatile("" : "+v"(blk_early[B].v[0]), "+v"(blk_early[B].v[1]), "+v"(blk_early[B].v[2]), "+v"(blk_early[B].v[3]), "+v"(tv));
      }
      __builtin_amdgcn_sched_barrier(0);              // (... and consumed behind them)
    }
    uint32_t probe = 0u;
#pragma unroll
    for (int i = 0; i < NLINES; ++i) probe |= pw[i];
    if ((probe == 0x9e3779b9u) & (blockIdx.x > 0x7ffffff0u)) atomicOr(&sc->err, 128u);
    tab = stage_log_table_finish(tv, tv1, tv2);
    PSTAMP(1);
  } else {
    tab = stage_log_table(logtab);
  }
  if (held & (4u | 8u | 16u | 64u)) return;
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[PropStage<D, IT>::bytes];
  PropAcc acc;
  propagate_range<D, false, IT, OBS, SUMS, ONE, WR>(src, src_stride, anc, dst, dst_stride, logw, gid0, seed, rec, mk, src2, 0, n_split, tab,
                                                range_lo, n, SUMS ? 1 : 0, SUMS ? pick_out : nullptr, pick_slot, s_stage, acc,
                                                nullptr, ONE ? &jp_early : nullptr, EARLY ? blk_early : nullptr, step_now, wsum_on);
  PSTAMP(6);
  if (!rec->has_obs) return;
  double tmax = wave_max(acc.tmax);
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = tmax;
  if (__any(acc.bad) && (threadIdx.x & 63) == 0) atomicOr(&sc->err, 1u);
  if (SUMS) {
    __shared__ cssm_u128 s_sa[CSSM_BLOCK / 64], s_sb[CSSM_BLOCK / 64];
    const cssm_u128 accS = wave_sum_u128(acc.S), accS2 = (SUMS == 2) ? wave_sum_u128(acc.S2) : cssm_u128_zero();
    if ((threadIdx.x & 63) == 0) {
      s_sa[threadIdx.x >> 6] = accS; if (SUMS == 2) s_sb[threadIdx.x >> 6] = accS2;
      if (wsum_on) subS2[(size_t)blockIdx.x * (CSSM_BLOCK / 64) + (threadIdx.x >> 6)] = accS;   // the quarter unit's exact sum (k_offspring_wave)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      cssm_u128 ta = s_sa[0], tb = (SUMS == 2) ? s_sb[0] : cssm_u128_zero();
#pragma unroll
      for (int w = 1; w < CSSM_BLOCK / 64; ++w) { ta = cssm_u128_add(ta, s_sa[w]); if (SUMS == 2) tb = cssm_u128_add(tb, s_sb[w]); }
      subS[blockIdx.x] = ta;
      if (SUMS == 2) subS2[blockIdx.x] = tb;
      if (grp_on) group_sums_add(sc, grp_set, blockIdx.x >> grp_shift, ta, tb, SUMS == 2);   // (uniform) Scalars::grp / grp2
    }
  } else {
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double m = s_max[0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) m = (s_max[w] > m) ? s_max[w] : m;
    atomicMax(&sc->maxslot[((size_t)slot_set * CSSM_MAXSLOTS + blockIdx.x % CSSM_MAXSLOTS) * CSSM_SLOT_STRIDE],
              (unsigned long long)cssm_order_key(m));
  }
  PSTAMP(7);
}

// MKW != 0 (d <= 12): the model's structure -- per component its SDE, its place in the f map, where its leaf ends: ModelK::comp[0..2]
// -- at COMPILE time.  The reference composes its models at compile time too; here the structure is data, and every
// transition and every term of f branches on it (wave-uniform scalar branches, ~60 per tile of a d = 3 model).  The structures
// of the reference's example models at d <= 4 have instantiations of their own (cssm_prop.hip: KnownStructures); the host
// launches one only when the handle's ModelK::comp[0] IS that word, so the constant is the value the argument holds anyway.
// Bench model: k_propagate 18.4 -> 17.7 us event-bracketed at N = 2^20.
template <int D, int IT, int OBS, int SUMS, int ONE = 0, uint32_t MKW = 0u, uint32_t MKW1 = 0u, uint32_t MKW2 = 0u>
__global__ __launch_bounds__(CSSM_BLOCK, (ONE ? 2 : PropWaves<D, SUMS>::value)) void k_propagate_self(
    const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc, double* __restrict__ dst, size_t dst_stride,
    double* __restrict__ logw, uint64_t n_arg, uint64_t seed, const StepRec* __restrict__ rec, ModelK mk, Scalars* __restrict__ sc,
    int slot_set, const double* __restrict__ logtab, uint64_t chunk,
    cssm_u128* __restrict__ subS, cssm_u128* __restrict__ subS2, double* __restrict__ pick_out, uint32_t pick_slot, uint32_t step_now) {
  // (step_now: the observation's index = rec->step, passed by the host so that the ONE instantiations can use it before any load lands)
  static_assert(MKW == 0u || D <= 12, "three structure words cover twelve components");
  if (MKW != 0u) { mk.comp[0] = MKW; mk.comp[1] = MKW1; mk.comp[2] = MKW2; }
  propagate_block<D, IT, OBS, SUMS, ONE>(src, src_stride, anc, dst, dst_stride, logw, n_arg, seed, rec, mk, sc, slot_set, logtab, chunk, subS, subS2,
                                         pick_out, pick_slot, 0ull, nullptr, 0u, step_now);
}

// B independent clouds in one launch (blockIdx.y = the chain; ChainBase, cssm_device.hip.h): the body of k_propagate_self with the
// fused sums, the chain's buffers, records, Philox key and -- for `filter` -- path taken from its ChainBase.  At N = 100 000 (the
// PMMH configuration) one chain's launch leaves three quarters of the GPU idle; four chains fill it for the price of one.
template <int D, int IT, int OBS, int ONE = 0, uint32_t MKW = 0u, uint32_t MKW1 = 0u, uint32_t MKW2 = 0u>
__global__ __launch_bounds__(CSSM_BLOCK, (ONE ? 2 : PropWaves<D, 1>::value)) void k_propagate_batch(
    const ChainBase* __restrict__ chains, int cur, int anc_valid, size_t stride, uint64_t n_arg, uint32_t rec_idx, ModelK mk, int slot_set,
    const double* __restrict__ logtab, uint64_t chunk, int want_pick, uint32_t step_now) {
  static_assert(MKW == 0u || D <= 12, "three structure words cover twelve components");
  if (MKW != 0u) { mk.comp[0] = MKW; mk.comp[1] = MKW1; mk.comp[2] = MKW2; }
  const ChainBase* __restrict__ c = chains + blockIdx.y;
  const StepRec* __restrict__ rec = c->recs + rec_idx;
  // (`filter`: the thread that gathers the slot sampleOne picked after the observation before records that state on the way)
  double* pick_out = (want_pick && rec_idx >= 1u) ? c->path + (size_t)rec_idx * D : nullptr;
  const uint32_t pick_slot = (want_pick && rec_idx >= 1u) ? c->recs[rec_idx - 1u].pick : 0u;
  propagate_block<D, IT, OBS, 1, ONE>(c->state[cur], stride, anc_valid ? c->anc : nullptr, c->state[cur ^ 1], stride, c->logw, n_arg, c->seed, rec, mk, c->sc,
                                      slot_set, logtab, chunk, c->tileS, c->tileS2, pick_out, pick_slot, 0ull, nullptr, 0u, step_now);
}

// The sharded filter's slim launch (single-collective exchange: the sums are always formed, the rows received from the two
// neighbouring ranks are read in place -- src2 = the receive buffer, rows of D + 1 doubles, indices >= n_split --, max-slot set
// 0, an even first global particle: whole pairs per thread): the same body behind 18 arguments instead of the generic
// kernel's 24.  ONE = 2: tile after tile (units of at most CSSM_LOOP_MAX_TILES tiles); ONE = 0: software-pipelined.
// MKW..: a known structure at compile time, as in k_propagate_self.
template <int D, int IT, int OBS, int ONE, uint32_t MKW = 0u, uint32_t MKW1 = 0u, uint32_t MKW2 = 0u>
__global__ __launch_bounds__(CSSM_BLOCK, (ONE ? 2 : PropWaves<D, 2>::value)) void k_propagate_shard(
    const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc, double* __restrict__ dst, size_t dst_stride,
    double* __restrict__ logw, uint64_t n_arg, uint64_t gid0, uint64_t seed, const StepRec* __restrict__ rec, ModelK mk,
    Scalars* __restrict__ sc, const double* __restrict__ src2, uint32_t n_split, const double* __restrict__ logtab, uint64_t chunk,
    cssm_u128* __restrict__ subS, cssm_u128* __restrict__ subS2, uint32_t step_now, int slot_set) {
  // slot_set: max-slot set 0 in bits 0-7; bit 8 / bits 9-10: group sums wanted (one block per unit) and their set (propagate_block)
  if (MKW != 0u) { mk.comp[0] = MKW; mk.comp[1] = MKW1; mk.comp[2] = MKW2; }
#ifndef CSSM_EXP_SHARD_SUMS
#define CSSM_EXP_SHARD_SUMS 2
#endif
  propagate_block<D, IT, OBS, CSSM_EXP_SHARD_SUMS, ONE>(src, src_stride, anc, dst, dst_stride, logw, n_arg, seed, rec, mk, sc, slot_set, logtab, chunk, subS, subS2,
                                      nullptr, 0u, gid0, src2, n_split, step_now);
}

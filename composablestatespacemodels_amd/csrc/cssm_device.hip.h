// cssm_device.hip.h -- gfx950 device-side building blocks of libcssm_pf shared by every translation unit: record and
// scalar layouts, DPP scans, the contract's variate streams, transitions, f, log-densities, LDS DMA, bulk stores.
// Only inline device functions and templates live here (safe to include from several .hip files); the kernels are in
// cssm_propagate.hip.h (k_propagate, one translation unit per latent dimension) and cssm_kernels.hip.h (all others).
//
// Data layout in HBM: particles are struct-of-arrays fp64,
// state[k * stride + i] = latent component k (Tree.flatten order) of particle i, two buffers
// that ping-pong; logw[N] fp64; anc[stride] u32; endslot[N] u32 (sharded filter only: exclusive end of the
// run of resampling slots particle j owns).  Resampling never moves particles: the NEXT propagate kernel
// reads its input through anc[] (fused gather), so a step costs one read and one write of the cloud.
//
// Kernels of one observation (element-wise / scan work, no MFMA):
//   k_propagate   gather + exact SDE transition (or Euler-Maruyama) + f + log-density, block max -> one integer
//                 atomicMax per block; the next tile's states are prefetched by asynchronous global -> LDS loads
//                 (global_load_lds_dwordx4), the indices of the tile after it into registers; optionally (SUMS)
//                 the fixed-point sums of exp(w - c) as well                       model/ParticleFilter.scala:118,123-124
//   k_tile_sums   w1 = exp(w - level) in 128-bit fixed point, one (S, S2) per unit of tiles                        :125
//   k_offspring   unit prefix + totals (every block sums the <= 1K unit sums itself), ll and ess (:127-128), per tile
//                 a DPP wave scan -> cumulative weight C_j -> end slot cnt(C_j); every particle writes its own run of
//                 slots into anc (single GPU), or the end slots are kept for the exchange   model/Resampling.scala:36-58,69
//   launch geometry (single GPU; docs/history/LABNOTES_rounds1-3.md, section 5c): k_propagate_self<..., ONE> has no software pipeline -- everything
//                 position-dependent requested in the first round of loads, the normals drawn while the gathered rows travel.
//                 Clouds below 2^20 particles: ONE = 1, one tile per block, one pair of sums per block (k_offspring totals up to
//                 4096 of them); from 2^20 on a block owns a whole unit of 1024 * k particles and runs the same body tile after
//                 tile (ONE = 2) while a unit has at most 8 tiles; beyond that the software-pipelined kernel (ONE = 0, d <= 3) or
//                 one tile per block again with k_reduce_units behind it (blocks' sums -> <= 1024 unit sums, d >= 4)
//   sharded only  k_boundary_pack + k_offspring_expand_spec (single-collective exchange); k_scan_tiles / k_global_sums,
//                 k_pack, k_expand (exact exchange: candidates -> slots)
// Cross-lane traffic is DPP, not ds_bpermute (5 vs 25 cycles per move on MI355X, tools/instr_rate.hip).
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "cssm_records.h"

#define CSSM_BLOCK 256
#ifndef CSSM_PROP_IT_MID
#define CSSM_PROP_IT_MID 2 /* particles per thread of k_propagate for d = 3 .. 8 (PropItems) */
#endif
#define CSSM_ITEMS 4
#define CSSM_TILE (CSSM_BLOCK * CSSM_ITEMS) /* 1024 particles per tile */

// Device scalars of a handle.
#define CSSM_MAXSLOTS 64
#define CSSM_SLOT_STRIDE 16 /* u64 words: one 128-byte line per slot */
#define CSSM_MAXSETS 3
#define CSSM_GRP_UNITS 32 /* units per group, layout 1: up to 32 groups x 32 units -- ONE wave of k_offspring holds the groups' sums and its own group's units */
#define CSSM_GRP_MAX 64   /* groups the arrays hold.  Layout 2 (single GPU, clouds of more than 1024 units: 2^23 particles and up): up to 64 groups
                             x 64 units = 4096 units of <= 4 tiles -- one wave totals the groups, a second one the block's own group */
#define CSSM_GRP_SMALL 32 /* groups of layout 1 */
#define CSSM_GRP_LIMB 56  /* bits of the low limb */
#define CSSM_GRP_MAX_UNIT (1u << 15) /* particles per unit: sum < 2^(15 + 96 + 1) = 2^112 = two 56-bit limbs */
struct Scalars {
  // Order keys of the running max log-weight, sharded over CSSM_MAXSLOTS cache lines, in three sets: weighted
  // observation s uses set s % 3; the kernel that resamples it (k_offspring) clears the other two, so no reset kernel exists.
  unsigned long long maxslot[CSSM_MAXSETS * CSSM_MAXSLOTS * CSSM_SLOT_STRIDE];
  // Sums of GROUPS of CSSM_GRP_UNITS consecutive units (single GPU, fused sums, one k_propagate block per unit), in the same three
  // sets: every k_propagate block adds the two 56-bit limbs of its unit sum (< 2^112: the host turns the group sums on for units of
  // at most CSSM_GRP_MAX_UNIT particles, every weight being at most exp(2^-20)) to two 64-bit words, each on a cache line of its own (non-returning atomics; 32 blocks x 2^56 cannot overflow a word;
  // integer sums: any order, the same bits).  k_offspring's blocks then read 32 group sums + the 32 unit sums of their own group
  // instead of all 1024 unit sums (16 KiB per block, 16 MiB per launch through the L2s: doubling that traffic cost the kernel
  // 1.35 us at N = 2^20).  Cleared with the max slots.  Index: ((set * 2 + limb) * CSSM_GRP_MAX + group) * CSSM_SLOT_STRIDE.
  unsigned long long grp[CSSM_MAXSETS * 2 * CSSM_GRP_MAX * CSSM_SLOT_STRIDE];
  // The same for the sums of SQUARED weights of a shard's units (k_propagate_shard / the sharded LGCP launch form both sums, SUMS = 2: the
  // exchange ships them in the segment headers).  A shard's max slots stay in set 0; its group sums rotate through the three sets by
  // weighted observation like the single GPU's (cssm_pf::wparity), and block 0 of the exchange's offspring blocks clears the two other sets.
  unsigned long long grp2[CSSM_MAXSETS * 2 * CSSM_GRP_MAX * CSSM_SLOT_STRIDE];
  uint32_t err;              // bit0: NaN log-weight, bit1: all weights zero / max not finite,
                             // bit2: the reference level was unusable and the sums must be formed again (host retries)
                             // bit3: (sharded) the exchange capacity did not cover some rank's slots at step fail_step
                             // bit4: (sharded, peer-written exchange) a rank's segment did not arrive within the wait bound
                             // (bit 5: unused)
                             // bit6: batch series on hold at fail_step: its reference level was ruled out, the host redoes its sums
  int32_t ess;
  uint32_t fail_step;        // first observation whose exchange did not fit (0xffffffff: none); see k_offspring_expand_spec
  uint32_t pad_;
  double gmax;               // decoded global max of this step
  double ref;                // level the weights of this step were rescaled by (cssm_ref_choose)
  double next_ref;           // LGCP (StepRec::predict): the level predicted for the NEXT weighted observation, cssm_ref_predict(gmax) -- NaN
                             // while no weighted observation has run since the cloud was drawn / handed back by a host resampler
  double ll;                 // accumulated log-likelihood
  cssm_u128 S_local, S2_local; // local fixed-point sums (this rank)
  cssm_u128 S_off;           // sum of the ranks before this one
  cssm_u128 S_tot, S2_tot;   // global sums
  // ESS pending (single GPU, weights kept in place of log-weights): the sum of squared weights of the last weighted
  // observation was formed by k_offspring's blocks -- one partial per block in s2[pend_buf][0 .. pend_n) -- and nobody has
  // totalled it yet: the NEXT weighted observation's publisher block does (and files the value under ess_t[pend_idx] if the
  // batch call of generation pend_gen is still the one running), or the host when a call ends (cssm_ess_of, same arithmetic).
  uint32_t pend, pend_buf, pend_n, pend_idx, pend_gen;
  uint32_t wait_code;        // (peer-written exchange, with err bit 4) WHICH wait gave up: bit 0 an offspring block's header words, 1 an expansion
                             //   block's, 2 a pack row block's, 3 the eager rows' flag, 4 the flag of the rows beyond them, 5 a header flag
                             //   (launches without group sums); bits 8-15: the rank whose word it was
  cssm_u128 pend_S;          // S_tot of that observation
  // (peer-written exchange) how long a reader polls for a peer's words before it gives up (err bit 4), in ticks of the constant 100 MHz
  // clock (s_memrealtime): wall-clock time, not an iteration count -- a rank that compiles a kernel, a loaded host or a debugger may be
  // seconds late without being dead.  Set with the scalars (reset_scalars) from the handle's CSSM_PEER_TIMEOUT_MS (default 30 s).
  unsigned long long peer_wait_ticks;
  // (measurement) the constant 100 MHz clock (s_memrealtime) when the first kernel of the running batch call started: the kernel that
  // brings the call's first record(s) stamps it, k_finish hands it to the host next to its own stamp -- the DEVICE time of the call,
  // first instruction to completion word, without an event packet on the queue (cssm_pf_last_device_us)
  unsigned long long t_first;
  unsigned long long t_last;    // k_finish's own stamp (valid in the host's mirror only)
  // (measurement, peer-written exchange) what ONE block of every exchange launch spent polling its peers, in ticks of the 100 MHz clock,
  // summed since the cloud was drawn: the first offspring block for every rank's header words, the first expansion block for the header
  // words and then for its neighbours' eager-rows flags; xwaits = exchanges counted.  One thread of one block adds per launch (launches of
  // a stream follow each other): plain read-modify-write.  cssm_pf_shard_wait_stats; bench.py: per_rank.header_wait_us / rows_wait_us --
  // at world 1 the time a block needs to see its OWN header block's words; across GPUs the link shows up here, not in kernels_us.
  unsigned long long hdr_wait_ticks, xhdr_wait_ticks, rows_wait_ticks, xwaits;
};

// BATCHED independent filters (cssm_batch.hip: B clouds of one model structure -- the chains of a PMMH run, a pilot grid of
// parameters -- advanced in lockstep by ONE launch per stage, blockIdx.y = the chain): what differs between the chains of a launch.
// The array is rewritten at the start of every batched call; everything else of a launch (sizes, parities, the record's index) is
// common to the chains and travels as kernel arguments.
struct ChainBase {
  double* state[2]; double* logw; uint32_t* anc; cssm_u128* tileS; cssm_u128* tileS2; Scalars* sc; StepRec* recs;
  cssm_u128* s2buf; double* ll_t; int32_t* ess_t; double* path; const double* m0; const double* sd0;
  Scalars* host_sc; double* host_ll_t; int32_t* host_ess_t; uint32_t* host_done;
  uint64_t seed; uint32_t gen, done_seq;
};

// ------------------------------------------------------------------------------------ helpers

// Cross-lane traffic goes through DPP (data-parallel primitives: a VALU move whose source lane is a fixed
// pattern), not through ds_bpermute (__shfl*): measured on MI355X (tools/instr_rate.hip) a DPP move issues in ~5
// cycles per wave, a ds_bpermute_b32 in ~25, and a 128-bit scan needs 24 of either.
//   row_shr:n   lane i of each row of 16 reads lane i-n of its row (lanes without a source keep `old` = 0)
//   row_bcast15 lane 15 of every row -> all lanes of the NEXT row (row_mask 0xa: rows 1 and 3 take it)
//   row_bcast31 lane 31 -> all lanes of rows 2 and 3 (row_mask 0xc)
// After the six steps lane i holds the inclusive prefix over lanes 0..i, lane 63 the wave total.
#define CSSM_DPP_ROW_SHR(n) (0x110 + (n))
#define CSSM_DPP_BCAST15 0x142
#define CSSM_DPP_BCAST31 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {   // lanes that receive nothing read 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp0_u64(uint64_t v) {
  return (uint64_t)dpp0<CTRL, ROW_MASK>((uint32_t)v) | ((uint64_t)dpp0<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ cssm_u128 dpp0_u128(cssm_u128 v) {
  cssm_u128 r;
  r.lo = dpp0_u64<CTRL, ROW_MASK>(v.lo);
  r.hi = dpp0_u64<CTRL, ROW_MASK>(v.hi);
  return r;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane) |
         ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32);
}

// inclusive scan across the 64 lanes (integer adds: any order gives the same bits)
// The DPP read is folded into the adds of the carry chain (v_add_co_u32_dpp / v_addc_co_u32_dpp: src0 comes from the other lane,
// a lane without a source adds 0; rows masked off by row_mask keep their value): 24 instructions per scan instead of the 48 of
// v_mov_b32_dpp + add (the compiler does not fold a DPP move into an add that is itself inline assembly, and without the
// assembly it does not keep the four limbs on one carry chain).  k_offspring scans once per tile of 1024 particles and once
// more per block: a tenth of its instructions at N = 2^20.  The leading s_nop covers the VALU-write -> DPP-read hazard
// (2 wait states) against whatever instruction the compiler placed before the block; inside it every DPP operand was written
// at least 3 instructions earlier.  All 64 lanes must be active.
#define CSSM_SCAN_STEP(CTRL)                                  \
  "v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL "\n\t"              \
  "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL "\n\t"        \
  "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc " CTRL "\n\t"        \
  "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " CTRL "\n\t"
__device__ __forceinline__ cssm_u128 wave_scan_u128(cssm_u128 v, int lane) {
  (void)lane;
  uint32_t a0 = (uint32_t)v.lo, a1 = (uint32_t)(v.lo >> 32), a2 = (uint32_t)v.hi, a3 = (uint32_t)(v.hi >> 32);
  asm volatile("s_nop 1\n\t"
               CSSM_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               CSSM_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               CSSM_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               CSSM_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
               CSSM_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
               CSSM_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 0"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
  cssm_u128 r;
  r.lo = (uint64_t)a0 | ((uint64_t)a1 << 32);
  r.hi = (uint64_t)a2 | ((uint64_t)a3 << 32);
  return r;
}
// off + (the value `inc` holds in the lane before; lane 0: 0): the exclusive prefix from an inclusive scan, one carry chain whose
// first operand is read through DPP wave_shr:1
__device__ __forceinline__ cssm_u128 wave_excl_add_u128(cssm_u128 inc, cssm_u128 off) {
  uint32_t a0 = (uint32_t)inc.lo, a1 = (uint32_t)(inc.lo >> 32), a2 = (uint32_t)inc.hi, a3 = (uint32_t)(inc.hi >> 32);
  const uint32_t b0 = (uint32_t)off.lo, b1 = (uint32_t)(off.lo >> 32), b2 = (uint32_t)off.hi, b3 = (uint32_t)(off.hi >> 32);
  uint32_t r0, r1, r2, r3;
  asm volatile("s_nop 1\n\t"
               "v_add_co_u32_dpp %0, vcc, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_addc_co_u32_dpp %1, vcc, %5, %9, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_addc_co_u32_dpp %2, vcc, %6, %10, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
               "v_addc_co_u32_dpp %3, vcc, %7, %11, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
  cssm_u128 r;
  r.lo = (uint64_t)r0 | ((uint64_t)r1 << 32);
  r.hi = (uint64_t)r2 | ((uint64_t)r3 << 32);
  return r;
}
// Loads of data another GPU (or another process on this one) wrote while this kernel runs -- the receive windows of the peer-written
// exchange, read behind their flags: system-scope loads (sc0 sc1) that no cache level of this GPU answers from a stale line.  NOT an
// acquire fence: a system-scope fence is an L2 invalidate per wave, and k_offspring_expand_spec's 1024 blocks queueing for it cost
// 34 us per launch (measured); the handful of loads per block that read a window bypass the caches instead.
__device__ __forceinline__ unsigned long long ld_sys_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double ld_sys_f64(const double* p) { return cssm_u2d(ld_sys_u64(reinterpret_cast<const unsigned long long*>(p))); }

// A wave-uniform value the VALU produced, moved into scalar registers (v_readfirstlane_b32): it stops occupying vector registers
// for as long as it lives.  k_offspring_self's tile loop keeps ~12 such words (the running prefix, N / S_tot, the distance bound).
__device__ __forceinline__ double uniform_f64(double x) {
  const uint64_t b = cssm_d2u(x);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  return cssm_u2d((uint64_t)lo | ((uint64_t)hi << 32));
}
__device__ __forceinline__ cssm_u128 uniform_u128(cssm_u128 v) {
  cssm_u128 r;
  r.lo = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v.lo) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v.lo >> 32)) << 32);
  r.hi = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v.hi) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v.hi >> 32)) << 32);
  return r;
}
// x - floor(x) of a non-negative finite x (v_fract_f64)
__device__ __forceinline__ double cssm_fract_pos(double x) { return __builtin_amdgcn_fract(x); }
// wave total, uniform (it is read from lane 63 into scalar registers)
__device__ __forceinline__ cssm_u128 wave_sum_u128(cssm_u128 v) {
  v = wave_scan_u128(v, 0);
  cssm_u128 r;
  r.lo = readlane_u64(v.lo, 63);
  r.hi = readlane_u64(v.hi, 63);
  return r;
}
// (a missing DPP source reads 0, the identity of an unsigned max: the compiler folds the read into v_max_u32_dpp)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {   // uniform
#define CSSM_MAX_STEP(CTRL, RM) { const uint32_t o = dpp0<CTRL, RM>(v); v = (o > v) ? o : v; }
  CSSM_MAX_STEP(CSSM_DPP_ROW_SHR(1), 0xf) CSSM_MAX_STEP(CSSM_DPP_ROW_SHR(2), 0xf) CSSM_MAX_STEP(CSSM_DPP_ROW_SHR(4), 0xf)
  CSSM_MAX_STEP(CSSM_DPP_ROW_SHR(8), 0xf) CSSM_MAX_STEP(CSSM_DPP_BCAST15, 0xa) CSSM_MAX_STEP(CSSM_DPP_BCAST31, 0xc)
#undef CSSM_MAX_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit keys as two 32-bit reductions -- the high words, then the low words of the lanes that hold the winning high word:
// 14 + 2 instructions instead of six rounds of two DPP moves, a 64-bit compare and two selects
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {   // uniform
  const uint32_t hi = (uint32_t)(k >> 32);
  const uint32_t mh = wave_max_u32(hi);
  const uint32_t ml = wave_max_u32((hi == mh) ? (uint32_t)k : 0u);
  return ((uint64_t)mh << 32) | (uint64_t)ml;
}
// max of doubles through their order-preserving keys (a missing DPP source reads key 0, below every real key)
__device__ __forceinline__ double wave_max(double v) { return cssm_order_unkey(wave_max_u64(cssm_order_key(v))); }

// The contract's log table (include/cssm_numerics.h, CSSM_LOG_TAB) staged in LDS by every kernel that
// draws normals: `tab_global` is the handle's device copy.  All threads of the block must call it.
// (contract v7: the table also holds the 256 (sin, cos) pairs of cssm_sincos_u24 -- 768 doubles, 6 KiB)
__device__ __forceinline__ const double* stage_log_table(const double* __restrict__ tab_global) {
  __shared__ double s_logtab[CSSM_TAB_SIZE];
  for (int i = threadIdx.x; i < CSSM_TAB_SIZE; i += blockDim.x) s_logtab[i] = tab_global[i];
  __syncthreads();
  return s_logtab;
}
// The same in two halves for blocks of exactly 256 threads (the small-cloud kernels): the load is issued by the caller, next to
// its other first loads (`v` = tab_global[threadIdx.x]), and lands in LDS here.
__device__ __forceinline__ const double* stage_log_table_finish(double v, double v1, double v2) {
  static_assert(CSSM_TAB_SIZE == 3 * 256, "three table entries per thread of a 256-thread block");
  __shared__ double s_logtab2[CSSM_TAB_SIZE];
  s_logtab2[threadIdx.x] = v; s_logtab2[256 + threadIdx.x] = v1; s_logtab2[512 + threadIdx.x] = v2;
  __syncthreads();
  return s_logtab2;
}

// Normals of a stream by blocks (include/cssm_numerics.h, counter layout, contract v6): one Philox block = two Box-Muller
// pairs = four normals; E[m] = normal m of the stream counted from block b0 on, NP pairs evaluated.
template <int NP>
__device__ __forceinline__ void stream_normals(uint64_t seed, uint64_t stream, uint32_t step, uint32_t tag, uint32_t b0, const double* tab,
                                               double (&E)[2 * NP]) {
#pragma unroll
  for (int B = 0; B < (NP + 1) / 2; ++B) {
    const cssm_u32x4 blk = cssm_philox_draw(seed, stream, step, tag, b0 + (uint32_t)B);
    cssm_normal_pair64(blk.v[0], blk.v[1], tab, &E[(4 * B) % (2 * NP)], &E[(4 * B + 1) % (2 * NP)]);
    if (2 * B + 1 < NP) cssm_normal_pair64(blk.v[2], blk.v[3], tab, &E[(4 * B + 2) % (2 * NP)], &E[(4 * B + 3) % (2 * NP)]);
  }
}

// The d standard normals of global particle gid for an ordinary step or the initial draw: particles 2m and 2m+1 share
// stream m, particle gid owns its normals q = (gid & 1) * D + k.  An even particle reads normals 0 .. D-1 from block 0 on; an
// odd one normals D .. 2D-1, i.e. from block D >> 2 on with an offset of D & 3 normals into it -- a compile-time shift, so the
// choice between the two is one select per component and neighbouring lanes do not diverge.
template <int D>
__device__ __forceinline__ void draw_normals(uint64_t seed, uint64_t gid, uint32_t step, uint32_t tag,
                                             const double* tab, double (&z)[D]) {
  constexpr int S = D & 3;                       // offset of an odd particle's first normal inside its first block
  constexpr int NP = (D + S + 1) / 2;            // pairs that cover normals 0 .. D - 1 + S
  const bool odd = (gid & 1u) != 0u;
  const uint32_t b0 = odd ? (uint32_t)(D >> 2) : 0u;
  double E[2 * NP];
  stream_normals<NP>(seed, cssm_pair_stream(gid), step, tag, b0, tab, E);
#pragma unroll
  for (int k = 0; k < D; ++k) z[k] = odd ? E[k + S] : E[k];
}

// One transition of component k (model/Sde.scala:86-95,114-123,139-150; :30-43 for Euler); components are independent.
template <int D>
__device__ __forceinline__ void transition_one(const ModelK& mk, const StepRec* __restrict__ rec, double dt, int k, double& xk, double zk) {
  const double p0 = rec->coef[k][0], p1 = rec->coef[k][1], p2 = rec->coef[k][2], p3 = rec->coef[k][3];
  const int kind = mk.kind(k);
  if (kind == CSSM_SDE_BROWNIAN) {
    xk = p3 * zk + xk;
  } else if (kind == CSSM_SDE_GEN_BROWNIAN) {
    double mean = xk + p0;
    xk = p3 * zk + mean;
  } else if (kind == CSSM_SDE_OU) {
    double mean = p0 + (xk - p0) * p1;
    xk = p3 * zk + mean;
  } else {
    double dW = p3 * zk;
    double a = (p0 + p1 * xk) * dt;
    double b = p2 * dW;
    xk = (xk + a) + b;
  }
}
template <int D>
__device__ __forceinline__ void transition(const ModelK& mk, const StepRec* __restrict__ rec, double dt,
                                           double (&x)[D], const double (&z)[D]) {
#pragma unroll
  for (int k = 0; k < D; ++k) transition_one<D>(mk, rec, dt, k, x[k], z[k]);
}

// Ordinary step of the two particles of a pair (2m, 2m+1): ceil(D / 2) Philox blocks give their 2 D normals, each fed to
// its component as soon as it exists (normal q -> particle q / D, component q % D).
template <int D>
__device__ __forceinline__ void propagate_pair(const ModelK& mk, const StepRec* __restrict__ rec, double dt, uint64_t seed,
                                               uint64_t gid_even, uint32_t step, const double* tab, double (&xa)[D], double (&xb)[D]) {
  const uint64_t stream = cssm_pair_stream(gid_even);
  auto feed = [&](int q, double e) {            // (q is a compile-time constant wherever this is called)
    if (q < D) transition_one<D>(mk, rec, dt, q, xa[q % D], e);
    else if (q < 2 * D) transition_one<D>(mk, rec, dt, q - D, xb[(q - D + D) % D], e);
  };
#pragma unroll
  for (int B = 0; B < (D + 1) / 2; ++B) {
    const cssm_u32x4 blk = cssm_philox_draw(seed, stream, step, CSSM_STREAM_STEP, (uint32_t)B);
    double e0, e1;
    cssm_normal_pair64(blk.v[0], blk.v[1], tab, &e0, &e1);
    feed(4 * B, e0); feed(4 * B + 1, e1);
    if (2 * B + 1 < D) {
      cssm_normal_pair64(blk.v[2], blk.v[3], tab, &e0, &e1);
      feed(4 * B + 2, e0); feed(4 * B + 3, e1);
    }
  }
}
// The pair's normals on their own (k_propagate's single-tile instantiations draw them while the gathered rows travel): half
// H = 0 / 1 of the pair's ceil(D / 2) Philox blocks, z[q] = normal q of the pair (-> particle q / D, component q % D, exactly as in
// propagate_pair).  PairHalf<D>::n0 normals come with half 0.
template <int D> struct PairHalf {
  static constexpr int nblk = (D + 1) / 2;                       // blocks of the pair
  static constexpr int b_split = nblk / 2;                       // half 0 = blocks [0, b_split), half 1 = [b_split, nblk)
  static constexpr int n0 = (4 * b_split < 2 * D) ? 4 * b_split : 2 * D;
};
template <int D, int H>
__device__ __forceinline__ void normals_pair_half(uint64_t seed, uint64_t gid_even, uint32_t step, const double* tab, double* z) {
  const uint64_t stream = cssm_pair_stream(gid_even);
  constexpr int B0 = H == 0 ? 0 : PairHalf<D>::b_split, B1 = H == 0 ? PairHalf<D>::b_split : PairHalf<D>::nblk;
#pragma unroll
  for (int B = B0; B < B1; ++B) {
    const cssm_u32x4 blk = cssm_philox_draw(seed, stream, step, CSSM_STREAM_STEP, (uint32_t)B);
    cssm_normal_pair64(blk.v[0], blk.v[1], tab, &z[4 * B], &z[4 * B + 1]);
    if (2 * B + 1 < D) cssm_normal_pair64(blk.v[2], blk.v[3], tab, &z[4 * B + 2], &z[4 * B + 3]);
  }
}
// ... and from Philox blocks drawn beforehand (blk[B] = block B of the pair's stream: k_propagate's single-tile instantiations draw
// the first tile's blocks while their first loads -- ancestor indices, the contract's table -- are still on their way)
template <int D>
__device__ __forceinline__ void normals_pair_from_blocks(const cssm_u32x4* blk, const double* tab, double* z) {
#pragma unroll
  for (int B = 0; B < PairHalf<D>::nblk; ++B) {
    cssm_normal_pair64(blk[B].v[0], blk[B].v[1], tab, &z[4 * B], &z[4 * B + 1]);
    if (2 * B + 1 < D) cssm_normal_pair64(blk[B].v[2], blk[B].v[3], tab, &z[4 * B + 2], &z[4 * B + 3]);
  }
}
// The same for ONE particle of either parity (threads that do not own whole pairs)
template <int D>
__device__ __forceinline__ void propagate_one(const ModelK& mk, const StepRec* __restrict__ rec, double dt, uint64_t seed,
                                              uint64_t gid, uint32_t step, const double* tab, double (&x)[D]) {
  double z[D];
  draw_normals<D>(seed, gid, step, CSSM_STREAM_STEP, tab, z);
  transition<D>(mk, rec, dt, x, z);
}

// gamma = f(x, t): per-leaf dot product, leaves summed left-nested (model/Model.scala:122-128,217-225,271)
template <int D>
__device__ __forceinline__ double gamma_coef(const ModelK& mk, const double* __restrict__ fco, const double (&x)[D]) {
  double g = 0.0, acc = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const int fm = mk.fmode(k);
    if (fm == FM_START) acc = fco[k] * x[k];
    else if (fm == FM_ADD) acc = acc + fco[k] * x[k];
    if (mk.leaf_end(k)) g = mk.first_leaf(k) ? acc : g + acc;
  }
  return g;
}
template <int D>
__device__ __forceinline__ double gamma_of(const ModelK& mk, const StepRec* __restrict__ rec, const double (&x)[D]) {
  return gamma_coef<D>(mk, rec->fco, x);
}

// dataLikelihood(gamma, y) of the leftmost leaf; the branch is wave-uniform (mk is a kernel argument).
// Constants c[] per observation kind are listed in build_rec (cssm_pf.hip); the oracle states the same
// expressions with the reference's line numbers (oracle/cssm_oracle.c, logdens).
// OBS >= 0: the observation kind is a compile-time constant (the common Poisson / Gaussian kernels carry only
// their own density: the generic body is ~2.5x larger and spills out of the instruction cache); OBS < 0: runtime.
template <int OBS>
__device__ __forceinline__ double logdens(const ModelK& mk, const StepRec* __restrict__ rec, double g, const double* tab) {
  const double y = rec->y;
  switch (OBS >= 0 ? OBS : mk.obs_kind) {
    case CSSM_OBS_POISSON:   // -lambda + k log(lambda) - lgamma(k+1), model/Model.scala:273
      return -cssm_exp(g) + y * g - rec->c[0];
    case CSSM_OBS_GAUSSIAN: {  // breeze Gaussian.logPdf, model/Model.scala:252-258
      const double dd = (y - g) / rec->c[1];
      return -(dd * dd) / 2.0 - rec->c[0];
    }
    case CSSM_OBS_NEGBIN: {    // model/Model.scala:186-195
      const double size = rec->c[1], mu = cssm_exp(g);
      return rec->c[0] + size * cssm_log(size / (mu + size)) + y * cssm_log(mu / (mu + size));
    }
    case CSSM_OBS_ZIP: {       // model/Model.scala:298-307
      if (y == 0.0) return cssm_log(rec->c[0] + (1.0 - rec->c[0]) * cssm_exp(-cssm_exp(g)));
      return ((rec->c[1] + y * g) - cssm_exp(g)) - rec->c[2];
    }
    case CSSM_OBS_BERNOULLI: { // model/Model.scala:318-336
      const double link = (g > 6.0) ? 1.0 : ((g < -6.0) ? 0.0 : 1.0 / (1.0 + cssm_exp(-g)));
      if (y == 1.0) return (link == 0.0) ? -1e99 : cssm_log(link);
      return (link == 1.0) ? -1e99 : cssm_log(1.0 - link);
    }
    case CSSM_OBS_STUDENT_T: { // 1/v * StudentsT(df).logPdf((y - eta)/v), model/Model.scala:155-160
      const double x = (y - g) / rec->c[1];
      return rec->c[3] * (rec->c[0] - rec->c[2] * cssm_log(1.0 + (x * x) / rec->cdf));
    }
    default: {                 // Beta(exp(-gamma), 1).logPdf(y) = (a - 1) log y + log a, model/Model.scala:349-352
      return (cssm_exp(-g) - 1.0) * rec->c[0] - g;
    }
  }
}


// ------------------------------------------------------------------------------------ propagate + weight

// Block-cooperative decode of the sharded running max: lane t of wave 0 reads slot t (one load
// latency instead of 64), wave max, broadcast through LDS.  All threads of the block must call it.
__device__ __forceinline__ double block_decode_slots(const Scalars* __restrict__ sc, int set) {
  __shared__ unsigned long long s_key;
  if (threadIdx.x < 64) {
    unsigned long long k = (threadIdx.x < CSSM_MAXSLOTS)
        ? sc->maxslot[((size_t)set * CSSM_MAXSLOTS + threadIdx.x) * CSSM_SLOT_STRIDE] : 0ull;
    k = wave_max_u64(k);
    if (threadIdx.x == 0) s_key = k;
  }
  __syncthreads();
  return cssm_order_unkey(s_key);
}

// Asynchronous 16-byte-per-lane copy global -> LDS (gfx950: global_load_lds_dwordx4): lane l fetches the 16 bytes at
// its own address g into LDS bytes [lds_base + 16 l, +16); lds_base is wave-uniform and travels in M0.  Issued through
// inline assembly ON PURPOSE: for the builtin form the compiler, unable to prove that later LDS reads (the log table)
// do not touch the destination, inserts s_waitcnt vmcnt(0) before the first LDS read that follows, i.e. it waits for
// the very prefetch that is meant to overlap the computation.  An operation the compiler does not count only makes its
// own counted waits more conservative (VMEM operations retire in order), never wrong; completion is awaited explicitly
// (s_waitcnt vmcnt(0)) before the wave reads the region back.  M0 has no other use in these kernels.
__device__ __forceinline__ void lds_dma16(const double* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory");
}
// The same with 4 bytes per lane (lane l -> LDS bytes [lds_base + 4 l, +4)): a double travels as two of these, low and
// high word into two 256-byte regions, where LDS is too small for 16 bytes per element.
__device__ __forceinline__ void lds_dma4(const void* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_base) : "memory");
}

// How k_propagate's two-particles-per-thread instantiations (d = 3 .. 8) store their bulk outputs (state rows,
// log-weights), one 16-byte store per lane and row, contiguous across the wave.  0: plain stores (dirty lines stay in
// the L2 and are written back when the kernel ends); 1: non-temporal; 2: write-through at agent scope (sc1); 3: sc0 sc1.
// Measured per step at N = 2^20 / 2^24 (d = 3, separate sums): 40.7 / 368 us plain, 39.8 / 374 nt, 37.9 / 366 sc1,
// 37.9 / 366 sc0 sc1 -- the write-back of a launch's last dirty lines is on its critical path, a write-through store is
// not (tools/sc1_stream_bench.hip shows the same on a bare stream).  Only where a wave's store instruction fills whole
// lines: the four-particles-per-thread layout (d <= 2: two 16-byte stores per lane, 32 bytes apart) loses the L2's
// write combining with sc1 (k_propagate<1> 129 -> 190 us at N = 2^24), so do the 4-byte run writes of k_offspring
// (106 -> 165 us); one particle per thread (d >= 9, 8-byte stores) is neutral (191.6 vs 192.1 us).  Those stay plain.
// HAZARD: a vector-memory store of more than 64 bits reads its data registers over several cycles, and a VALU instruction
// that overwrites them in the next wait states corrupts what is stored (gfx9 "VMEM store data hazard").  The compiler
// pads its own stores; it cannot see into inline assembly, so every 128-bit store issued from inline assembly carries
// its own `s_nop 1` (found the hard way: ancestor indices came out as halves of the NEXT store's address in exactly
// those instantiations whose schedule happened to put the address arithmetic right behind the store).
#ifndef CSSM_ST_MODE
#define CSSM_ST_MODE 2
#endif
typedef double cssm_dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bulk_store2(double* p, double a, double b) {
#if CSSM_ST_MODE == 1
  cssm_dbl2 v; v.x = a; v.y = b;
  __builtin_nontemporal_store(v, reinterpret_cast<cssm_dbl2*>(p));
#elif CSSM_ST_MODE == 2
  cssm_dbl2 v; v.x = a; v.y = b;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");   // s_nop: see CSSM_ST_MODE
#elif CSSM_ST_MODE == 3
  cssm_dbl2 v; v.x = a; v.y = b;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
#endif
}

static_assert(CSSM_GRP_UNITS == 32, "group index = block >> (5 + log2(blocks per unit))");
// A unit's sum (and, squares: its sum of squared weights) added to its group's: two 56-bit limbs each, non-returning atomics on lines of
// their own (Scalars::grp / grp2; integer sums: any order, the same bits).  One thread of the unit's block calls it.
__device__ __forceinline__ void group_sums_add(Scalars* __restrict__ sc, int set, uint32_t group, cssm_u128 ta, cssm_u128 tb, bool squares) {
  const size_t at = ((size_t)set * 2 * CSSM_GRP_MAX + group) * CSSM_SLOT_STRIDE;
  unsigned long long* g = &sc->grp[at];
  atomicAdd(g, ta.lo & ((1ull << CSSM_GRP_LIMB) - 1ull));
  atomicAdd(g + (size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE, (ta.lo >> CSSM_GRP_LIMB) | (ta.hi << (64 - CSSM_GRP_LIMB)));
  if (squares) {
    unsigned long long* g2 = &sc->grp2[at];
    atomicAdd(g2, tb.lo & ((1ull << CSSM_GRP_LIMB) - 1ull));
    atomicAdd(g2 + (size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE, (tb.lo >> CSSM_GRP_LIMB) | (tb.hi << (64 - CSSM_GRP_LIMB)));
  }
}

// ess = floor(1 / sum (w1/tot)^2) (model/ParticleFilter.scala:128, :431-434) from the fixed-point sums; IEEE operations only:
// the host evaluates the same function when it totals a pending observation's squares itself
__host__ __device__ __forceinline__ int32_t cssm_ess_of(cssm_u128 S, cssm_u128 S2) {
  const double tot = cssm_fix_to_double(S);
  const double tot2 = cssm_fix_to_double(S2);
  const double e = 1.0 / (tot2 / (tot * tot));
  const double fl = (double)(long long)e;   // e >= 1 here; floor == trunc
  return (e < 2147483647.0) ? (int32_t)fl : 2147483647;
}
// The kernel that publishes a weighted observation's scalars also hands the NEXT weighted observation its level where that is
// predicted from this one's max (LGCP, contract v8): into the record behind this one -- the records of a call are consecutive,
// and the buffer holds one spare record behind the last -- and into Scalars::next_ref, from where the host chains the first
// record of the next call (k_chain_level).  One thread calls it.
__device__ __forceinline__ void publish_next_level(Scalars* __restrict__ sc, const StepRec* __restrict__ rec, double gmax_dec) {
  if (rec->predict) {
    const double c = cssm_ref_predict(gmax_dec);
    sc->next_ref = c;
    const_cast<StepRec*>(rec + 1)->ref = c;
  }
}
// ll += level + log(mean(w1)) (:127, :522-524) from sc->S_tot; false (and err bit 1) when no weight is left
__device__ __forceinline__ bool finish_ll(Scalars* sc, uint64_t n_global) {
  if (cssm_u128_is_zero(sc->S_tot) || !(sc->gmax > -cssm_inf()) || !(sc->gmax < cssm_inf())) {
    atomicOr(&sc->err, 2u);
    return false;
  }
  sc->ll = sc->ll + sc->ref + cssm_log(cssm_fix_to_double(sc->S_tot) / (double)n_global);
  return true;
}
// ... and the ESS with it, when the sum of squares is at hand (sc->S2_tot)
__device__ __forceinline__ void finish_step(Scalars* sc, uint64_t n_global) {
  if (finish_ll(sc, n_global)) sc->ess = cssm_ess_of(sc->S_tot, sc->S2_tot);
}


// ------------------------------------------------------------------------------------ weights of a tile, end slots

// raw != 0: `logw` already holds the weights w1 themselves -- 1: the stateless Resample[A] entry point, whose second argument is
// w1 = exp(w - max) (model/ParticleFilter.scala:125-126; arbitrary host doubles); 2: k_propagate<SUMS> stored
// exp(min(w - c, 2^-20)) in place of the log-weights (in [0, 1 + 2^-20]: cssm_fix_from_unit applies).
__device__ __forceinline__ void load_tile_raw(const double* __restrict__ logw, uint64_t base, uint64_t n, int raw,
                                              double (&v)[CSSM_ITEMS]) {
  const uint64_t i0 = base + (uint64_t)threadIdx.x * CSSM_ITEMS;
  if (i0 + CSSM_ITEMS <= n) {
    const double2 a = *reinterpret_cast<const double2*>(logw + i0);
    const double2 b = *reinterpret_cast<const double2*>(logw + i0 + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) v[r] = (i0 + r < n) ? logw[i0 + r] : (raw ? 0.0 : -cssm_inf());
  }
}
__device__ __forceinline__ void weights_from_raw(const double (&v)[CSSM_ITEMS], double gmax, int raw, double (&w1)[CSSM_ITEMS],
                                                 const double* tab) {
#pragma unroll
  for (int r = 0; r < CSSM_ITEMS; ++r) w1[r] = raw ? v[r] : cssm_exp_le0(v[r] - gmax);   // (w - level <= 2^-20, never NaN: k_propagate)
}
__device__ __forceinline__ void load_tile_weights(const double* __restrict__ logw, uint64_t base, uint64_t n,
                                                  double gmax, int raw, double (&w1)[CSSM_ITEMS], const double* tab) {
  double v[CSSM_ITEMS];
  load_tile_raw(logw, base, n, raw, v);
  weights_from_raw(v, gmax, raw, w1, tab);
}

// inclusive max-scan across the 64 lanes (values >= 0; a lane without a DPP source reads 0)
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) {
#define CSSM_MX(CTRL, RM) { const uint32_t o = dpp0<CTRL, RM>(v); v = (o > v) ? o : v; }
  CSSM_MX(CSSM_DPP_ROW_SHR(1), 0xf) CSSM_MX(CSSM_DPP_ROW_SHR(2), 0xf) CSSM_MX(CSSM_DPP_ROW_SHR(4), 0xf)
  CSSM_MX(CSSM_DPP_ROW_SHR(8), 0xf) CSSM_MX(CSSM_DPP_BCAST15, 0xa) CSSM_MX(CSSM_DPP_BCAST31, 0xc)
#undef CSSM_MX
  return v;
}

typedef uint32_t cssm_u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_anc4_sc1(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  cssm_u32x4v v; v.x = a; v.y = b; v.z = c; v.w = d;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");   // s_nop: store-data hazard, see CSSM_ST_MODE
}


#define CSSM_WAVE_CHUNK 512   /* resampling slots one WAVE assembles in LDS at a time (2 KiB per wave) */

// findAllInTreeMap (model/Resampling.scala:36-46) for ONE WAVE's 256 particles, with no block barrier at all: lane l holds the
// end slots e[0..3] of its four consecutive particles first_idx .. first_idx + 3 and the end slot `prev` of the particle before
// them; the wave's particles own the slots [lo, hi) (wave-uniform; already clipped to what this launch may write).  The ancestor
// indices of those slots are assembled in LDS, 512 at a time -- every particle drops its index at the first slot of its run,
// an inclusive max-scan fills the runs (indices grow with the slots) -- and leave as whole lines (round 1 wrote 4 bytes per slot:
// 1.38x the algorithmic write traffic).  SC1: write-through stores.  anc is indexed by slot - slot_off (signed arithmetic: a
// chunk starts on a 64-slot boundary at or below lo, which can lie below slot_off; only slots inside [lo, hi) are dereferenced);
// idx_max: unused since the markers are the indices themselves (it clamped index + 1 - 1).  A wave's LDS operations execute in program order, so markers written by some lanes are visible to
// the reads of others without s_barrier -- the compiler is held to that order by wave_barrier + an explicit lgkmcnt wait.
// k_offspring spent half its wave cycles waiting (PMC SQ_WAIT_ANY 51 %): eight block barriers per tile of 1024 particles,
// five of them in the block-wide version of this function.  s_wave: CSSM_WAVE_CHUNK words of LDS owned by this wave.
template <bool SC1, bool CLIP = true>
__device__ __forceinline__ void fill_runs_wave(uint32_t prev, const uint32_t (&e)[CSSM_ITEMS], uint32_t first_idx, uint32_t lo, uint32_t hi,
                                               uint32_t* __restrict__ anc, uint32_t slot_off, uint32_t idx_max, uint32_t* __restrict__ s_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  (void)idx_max;
  auto lds_order = [] { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  for (uint32_t c0 = lo & ~63u; c0 < hi; c0 += CSSM_WAVE_CHUNK) {
    uint4* z = reinterpret_cast<uint4*>(s_wave + lane * 8);
    z[0] = make_uint4(0u, 0u, 0u, 0u); z[1] = make_uint4(0u, 0u, 0u, 0u);
    lds_order();
    if (!CLIP && c0 <= lo && hi - c0 <= CSSM_WAVE_CHUNK) {
      // (uniform) !CLIP: [lo, hi) is exactly the union of this wave's runs, and here all of it lies in this chunk: no run
      // needs clipping -- the common case (a wave's 256 particles own 256 slots on average, the chunk holds 512)
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) {
        const uint32_t rb = (r == 0) ? prev : e[r - 1];
        if (e[r] > rb) s_wave[rb - c0] = first_idx + r;
      }
    } else {
#pragma unroll
      for (int r = 0; r < CSSM_ITEMS; ++r) {
        uint32_t rb = (r == 0) ? prev : e[r - 1];
        uint32_t re = e[r];
        rb = (rb < lo) ? lo : rb;
        re = (re > hi) ? hi : re;
        if (re > rb && re > c0 && rb < c0 + CSSM_WAVE_CHUNK) {
          const uint32_t pos = ((rb > c0) ? rb : c0) - c0;
          s_wave[pos] = first_idx + r;                       // 0 = no run starts here, or particle 0's (see below)
        }
      }
    }
    lds_order();
    const uint4 a = z[0], bq = z[1];
    uint32_t v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = (v[k - 1] > v[k]) ? v[k - 1] : v[k];
    const uint32_t incl = wave_scan_max_u32(v[7]);
    const uint32_t carry = dpp0<0x138 /* wave_shr:1 */, 0xf>(incl);   // exclusive: max over the lanes before this one (lane 0: 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // (a marker is the index of the particle whose run starts at the slot; an empty slot reads 0, which is also particle 0's
      //  marker: the slots of [lo, hi) before every other marker can only be particle 0's run -- the cloud's first slots -- so the
      //  running max is the ancestor as it stands.  Every marker is a particle index of this tile: <= idx_max.)
      v[k] = (v[k] > carry) ? v[k] : carry;
    }
    // back through LDS so that each of the two store instructions of the wave covers 1 KiB contiguously
    lds_order();
    z[0] = make_uint4(v[0], v[1], v[2], v[3]); z[1] = make_uint4(v[4], v[5], v[6], v[7]);
    lds_order();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint32_t p = half * (CSSM_WAVE_CHUNK / 2) + lane * 4;
      const uint4 w = *reinterpret_cast<const uint4*>(s_wave + p);
      const uint32_t s0 = c0 + p;
      uint32_t* dst = anc + ((long long)s0 - (long long)slot_off);
      if (s0 >= lo && s0 + 4 <= hi && ((s0 - slot_off) & 3u) == 0u) {
        if (SC1) store_anc4_sc1(dst, w.x, w.y, w.z, w.w);
        else *reinterpret_cast<uint4*>(dst) = w;
      } else {
        const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (s0 + k >= lo && s0 + k < hi) {
            if (SC1) __hip_atomic_store(dst + k, wv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[k] = wv[k];
          }
      }
    }
    lds_order();                                            // the region is rewritten by the next chunk
  }
}

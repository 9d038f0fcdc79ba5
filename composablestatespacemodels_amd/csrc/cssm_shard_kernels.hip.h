// cssm_shard_kernels.hip.h -- kernels only the sharded filter launches (included once, by cssm_shard.hip): the
// single-collective exchange (k_boundary_pack, k_offspring_expand_spec) and the exact exchange (k_expand; k_send_ranges, k_pack, k_adopt_remote in cssm_shard.hip).
#pragma once

#include "cssm_kernels.hip.h"

// ------------------------------------------------------------------------------------ expand

// findAllInTreeMap (model/Resampling.scala:36-46) on the receiver of the sharded filter, for the slots of this rank
// [slot_lo, slot_hi) that belong to OTHER ranks' particles (the own particles wrote their runs in k_offspring).
// `cand_end` holds the end slots of the m received candidates in global particle order -- n_low from lower ranks,
// then those from higher ranks -- and `cand_idx` where each candidate's state lives.  Candidate j writes its run
// [max(start_j, slot_lo), min(end_j, slot_hi)) <- cand_idx[j], start_j = end_{j-1}; the first candidate from below
// starts at or before slot_lo by construction, the first one from above at the own last end slot.  Runs longer
// than CSSM_RUN_DIRECT are written by the whole block.
__device__ __forceinline__ void expand_body(const uint32_t* __restrict__ cand_end, const uint32_t* __restrict__ cand_idx,
                                            uint64_t m, uint64_t n_low, uint64_t slot_lo, uint64_t slot_hi,
                                            uint32_t* __restrict__ anc, const uint32_t* __restrict__ own_last_end) {
  __shared__ uint32_t s_nheavy;
  __shared__ uint32_t s_hb[CSSM_BLOCK], s_he[CSSM_BLOCK], s_hj[CSSM_BLOCK];
  for (uint64_t base = (uint64_t)blockIdx.x * CSSM_BLOCK; base < m; base += (uint64_t)gridDim.x * CSSM_BLOCK) {
    if (threadIdx.x == 0) s_nheavy = 0;
    __syncthreads();
    const uint64_t j = base + threadIdx.x;
    if (j < m) {
      uint64_t b = (j == n_low) ? (uint64_t)*own_last_end : ((j == 0) ? slot_lo : (uint64_t)cand_end[j - 1]);
      uint64_t e = cand_end[j];
      if (b < slot_lo) b = slot_lo;
      if (e > slot_hi) e = slot_hi;
      if (e > b) {
        const uint32_t idx = cand_idx[j];
        if (e - b <= CSSM_RUN_DIRECT) {
          for (uint64_t s = b; s < e; ++s) anc[s - slot_lo] = idx;
        } else {
          const uint32_t h = atomicAdd(&s_nheavy, 1u);
          s_hb[h] = (uint32_t)(b - slot_lo); s_he[h] = (uint32_t)(e - slot_lo); s_hj[h] = idx;
        }
      }
    }
    __syncthreads();
    const uint32_t nh = s_nheavy;
    for (uint32_t h = 0; h < nh; ++h) {
      const uint32_t he = s_he[h], hj = s_hj[h];
      for (uint32_t s = s_hb[h] + threadIdx.x; s < he; s += CSSM_BLOCK) anc[s] = hj;
    }
    __syncthreads();
  }
}
// ------------------------------------------------------------------------------------ single-collective exchange
//
// One all-to-all per observation carries BOTH the rank's 5 sum words and its boundary particles:
// segment r -> q of R-double rows (R = d + 1), laid out as
//   [0, HD)                header: [0] row count, [1..5] S.lo S.hi S2.lo S2.hi max-key (raw bits), [6..7] base (u128 raw bits),
//                          [8..9] total weight of the rank's FIRST-cap block, [10..11] of its LAST-cap block (every header carries
//                          both: with them EVERY rank can tell from the headers alone whether EVERY rank's slots are covered)
//   [HD, HD + cap R)       rows: (state d, low word of P_j)      P_j = inclusive prefix of the fixed-point weights
//   [HD + cap R, + capP)   high words of P_j                            WITHIN the block of particles the segment carries
// q = r - 1 receives the rank's FIRST cap particles (base = 0), q = r + 1 its LAST cap particles (base = S_local - P_total),
// every other q (q = r included: the all-to-all's own segment delivers every rank its own sums too) the header only.  The receiver knows all sums after
// the exchange and turns base + P_j into global cumulative weights and end slots itself (k_expand_spec).
constexpr int kSpecHeaderWords = 12;   // what a header-only segment carries
__host__ __device__ __forceinline__ long long spec_hdr(int d) { return (long long)(d + 1) * ((12 + d) / (d + 1)); }
__host__ __device__ __forceinline__ long long spec_capP(int d, long long cap) { return (long long)(d + 1) * ((cap + d) / (d + 1)); }
__host__ __device__ __forceinline__ long long spec_seg(int d, long long cap) { return spec_hdr(d) + cap * (d + 1) + spec_capP(d, cap); }

// PEER-WRITTEN exchange: instead of filling a send buffer for an all-to-all, the segments are written
// straight into the DESTINATION ranks' receive windows -- peer-mapped device memory (hipIpcOpenMemHandle across processes; plain
// pointers where the shards share a process) -- and a flag per (window, source rank) tells the destination's
// k_offspring_expand_spec that the segment is complete: no collective launch per observation.  Two windows alternate by exchange
// number (the rows of exchange e are gathered from by the propagate of the NEXT observation while exchange e + 1 is written).
// Protocol, per destination q, two flags per source rank: [0] the HEADER is complete -- the one block that writes it makes its stores
// visible at SYSTEM scope (__threadfence_system by every thread, then the block barrier) and stores the exchange number with
// system-scope release: everything a reader needs before it can resample its own particles, and the only flag on its critical
// path; [1] the ROWS are complete -- every block that wrote rows does the same fence, takes a ticket on a LOCAL counter, and the
// block that takes the last one stores the number: read only where the received rows are expanded, behind the reader's own
// particles, by when it has long been set.  The reader polls with system-scope loads, bounded, and
// reads the window with system-scope loads as well (ld_sys: no fence -- an L2 invalidate per wave of a 1024-block kernel cost 34 us).
struct PeerTable {
  double* win[2][64];              // win[p][q]: base of rank q's receive window p (world segments; segment r = what rank r wrote)
  unsigned int* flag[2][64];       // flag[p][q]: rank q's flags of window p, two 64-byte lines per source rank: header, rows
};
#define CSSM_PEER_FLAG_STRIDE 96   /* uint32 words between the flags of consecutive source ranks (header flag at 0, rows flag at 16, the header itself at 32) */
#define CSSM_PEER_FLAG_ROWS 16
// The header a SECOND time, as 24 self-validating 8-byte words behind the flags (word w = exchange number << 32 | half w of the 12 header
// doubles): an 8-byte store is atomic, so a reader that sees the exchange number in a word holds its data -- no flag to wait for first and
// no release in front of it (the header flag's system-scope release is an L2 write-back of the writing XCD).  The group-sum launches
// read these (one round trip from "the peer's header block has stored" to "this block has the header" instead of two); the flag and the
// header in the window stay for the readers that do not (small shards).
#define CSSM_PEER_FLAG_LL 32
#define CSSM_PEER_LL_WORDS 24
// A reader polls until the word it waits for holds the exchange number or `ticks` of the constant 100 MHz clock have gone by (wall-clock
// time: Scalars::peer_wait_ticks); the clock is read every 64th poll
__device__ __forceinline__ bool peer_poll_u32(const unsigned int* f, unsigned int want, unsigned long long ticks) {
  unsigned long long t0 = 0ull;
  unsigned int polls = 0u;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
    if ((++polls & 63u) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0ull) t0 = now; else if (now - t0 > ticks) return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
  return true;
}
#ifndef CSSM_POLL_LL_SLEEP
#define CSSM_POLL_LL_SLEEP 2   /* s_sleep units (64 clocks) between two polls of a header word (tools/archive/poll_sleep_experiment.sh: 2 and 8 are level) */
#endif
__device__ __forceinline__ bool peer_poll_ll(const unsigned long long* f, unsigned int want, unsigned long long ticks, unsigned int& half) {
  unsigned long long t0 = 0ull, v;
  unsigned int polls = 0u;
  while ((unsigned int)((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != want) {
    if ((++polls & 63u) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0ull) t0 = now; else if (now - t0 > ticks) return false;
    }
    __builtin_amdgcn_s_sleep(CSSM_POLL_LL_SLEEP);
  }
  half = (unsigned int)v;
  return true;
}

// NEEDED ROWS.  A boundary block's capacity is sized for the worst observation (6 sqrt(N_global) rows: 17 408 at 8 x 2^20, 696 KB per
// neighbour) while the neighbour's slots are owned by a few hundred of them on a typical one (median 660) -- and every row is a remote store
// over one xGMI link.  With all headers at hand the SENDER knows which of its rows the neighbour's slots need: row i of the block for rank + 1
// iff its end slot lies beyond this rank's last slot (a suffix of the block: end slots grow with i), row i of the block for rank - 1 iff its
// run starts below this rank's first slot (a prefix).  The row blocks of the pack therefore wait for the headers too (the header blocks wait
// for nothing, so nobody waits in a circle), write only those rows, and the block that takes the last ticket publishes their number in the
// word behind the rows flag (`need`: rows [cnt - need, cnt) of the block for rank + 1, rows [0, need) of the block for rank - 1) before it
// sets the flag.  The reader expands exactly those.
// ... and EAGER ROWS, so that none of this lies on the reader's critical path: the `eager` rows next to the boundary (four tiles = 4096 rows by default,
// CSSM_PEER_EAGER_ROWS; a typical observation needs a quarter of that) are written AT ONCE -- they need no header -- with a flag of their
// own (ROWS), long set by the time the reader has resampled its own particles.  Only rows beyond them go the way described above, behind
// a second flag (EXTRA), and the reader waits for that one only if the eager rows do not reach its first / last slot -- which it sees from
// the eager rows themselves (the cumulative weight in front of the first eager row travels with them: PSTART).  eager >= cap is "every
// row travels" (CSSM_PEER_ALL_ROWS=1, and what the collective exchange does); eager = 0 "needed rows only, all behind the headers".
#define CSSM_PEER_FLAG_NEED (CSSM_PEER_FLAG_ROWS + 1)    /* rows of the block that were needed (0: none beyond the eager ones) */
#define CSSM_PEER_FLAG_EXTRA (CSSM_PEER_FLAG_ROWS + 2)   /* exchange number: the needed rows beyond the eager ones are complete */
#define CSSM_PEER_FLAG_PSTART (CSSM_PEER_FLAG_ROWS + 4)  /* two 8-byte words: cumulative weight in front of the first eager row (block for rank + 1) */
// the handle's local words of the protocol (cssm_pf::peer_tickets, CSSM_PEER_TICKET_WORDS uint32): [0, 64) tickets per destination,
// [64] the flag of the unit-sum prefixes (merged kernel), [96, 100) results of the handshake, [100, 108) PackNeed::stat, [128, 192) PackNeed::need,
// [192, 256) tickets of the rows beyond the eager ones
#define CSSM_PEER_TICKET_WORDS 256
#define CSSM_PEER_TICKET_HDR_DONE 66   /* merged launch: header blocks that are done reading the max slots (the first offspring block clears them) */
#define CSSM_PEER_TICKET_STAT 100
#define CSSM_PEER_TICKET_NEED 128
#define CSSM_PEER_TICKET_EXTRA 192
struct SpecHeaders;
struct PackNeed {
  SpecHeaders* H; uint32_t* ll;          // LDS of the launch: the ranks' headers, the 24 halves per rank they arrive as
  const unsigned int* my_flags;          // this rank's flags of the window of this exchange (the peers' headers land there)
  unsigned int* need;                    // [q]: rows destination q needs -- max over the row blocks (device memory, zero between launches)
  unsigned long long* stat;              // [0] += rows written for the neighbours, [1] += neighbour segments, [2] += segments that needed rows
                                         //   beyond the eager ones (diagnostics)
  uint64_t n_global, seed, slot_lo, slot_hi;
  long long eager;                       // rows next to the boundary that travel at once (>= cap: all of them)
  int rs;
};

// What every block that needs the ranks' sums does first: the headers of all segments -> per-rank sums, offsets, block totals (and, in
// the offspring blocks, the verdict "every rank's slots are covered by its own particles plus its neighbours' boundary blocks").
struct SpecHeaders {
  cssm_u128 S[64], off[64], base[64], plow[64], phigh[64];
  long long cnt[64];
  unsigned long long cnts[64][4];
  cssm_u128 tot, tot2;
  unsigned long long key[64], gkey;      // the ranks' max keys, the largest of them
  cssm_u128 S2[64];
  int all_ok;
};
// (the header words of rank threadIdx.x, requested by the kernel together with everything else it starts from: the verdict
//  used to begin with three round trips one behind the other -- the sticky bits, the max keys, the headers -- and a fourth
//  for the record's u in its middle: 3 us before a block had so much as asked for its weights)
struct SpecHdrRegs { double w[12]; };
__device__ __forceinline__ SpecHdrRegs spec_load_headers(const double* __restrict__ recv, int world, long long cap, int d) {
  SpecHdrRegs g;
#pragma unroll
  for (int k = 0; k < 12; ++k) g.w[k] = 0.0;
  if ((int)threadIdx.x < world) {
    const double* h = recv + (size_t)threadIdx.x * spec_seg(d, cap);
    g.w[0] = ld_sys_f64(h); g.w[1] = ld_sys_f64(h + 1); g.w[2] = ld_sys_f64(h + 2); g.w[3] = ld_sys_f64(h + 6); g.w[4] = ld_sys_f64(h + 7);
    g.w[5] = ld_sys_f64(h + 8); g.w[6] = ld_sys_f64(h + 9); g.w[7] = ld_sys_f64(h + 10); g.w[8] = ld_sys_f64(h + 11);
    g.w[9] = ld_sys_f64(h + 3); g.w[10] = ld_sys_f64(h + 4); g.w[11] = ld_sys_f64(h + 5);   // S2, the key of the rank's max
  }
  return g;
}
// stage 1: the headers into LDS, the ranks' offsets, the totals and the largest max key (all threads call; one barrier)
__device__ __forceinline__ void spec_store_headers(SpecHeaders& H, const SpecHdrRegs& g, int world, long long cap) {
  if (threadIdx.x < 64) {
    cssm_u128 S = cssm_u128_zero(), bs = cssm_u128_zero(), pl = cssm_u128_zero(), ph = cssm_u128_zero();
    long long c = 0;
    if ((int)threadIdx.x < world) {
      S.lo = cssm_d2u(g.w[1]); S.hi = cssm_d2u(g.w[2]); bs.lo = cssm_d2u(g.w[3]); bs.hi = cssm_d2u(g.w[4]);
      pl.lo = cssm_d2u(g.w[5]); pl.hi = cssm_d2u(g.w[6]); ph.lo = cssm_d2u(g.w[7]); ph.hi = cssm_d2u(g.w[8]);
      c = (long long)g.w[0];
      c = (c < 0) ? 0 : ((c > cap) ? cap : c);
    }
    H.S[threadIdx.x] = S; H.base[threadIdx.x] = bs; H.plow[threadIdx.x] = pl; H.phigh[threadIdx.x] = ph; H.cnt[threadIdx.x] = c;
    cssm_u128 S2; S2.lo = cssm_d2u(g.w[9]); S2.hi = cssm_d2u(g.w[10]);
    const unsigned long long key = ((int)threadIdx.x < world) ? cssm_d2u(g.w[11]) : 0ull;
    H.S2[threadIdx.x] = S2; H.key[threadIdx.x] = key;
    // the ranks' offsets, the totals and the largest key by the wave's scans, out of the registers (the ranks beyond `world` hold zeros).
    // (Thread 0 used to walk the ranks through LDS behind a barrier: 1.4 us from "headers seen" to "headers in LDS" at world 8 against 0.8
    //  at world 1, in every block of the launch -- tools/archive/exchange_stamps_local.py)
    const int lane = (int)threadIdx.x;
    const cssm_u128 inc = wave_scan_u128(S, lane);
    cssm_u128 off; off.lo = inc.lo - S.lo; off.hi = inc.hi - S.hi - (inc.lo < S.lo ? 1ull : 0ull);
    H.off[threadIdx.x] = off;
    const cssm_u128 tot2 = wave_sum_u128(S2);
    const unsigned long long gkey = wave_max_u64(key);
    if (lane == 63) H.tot = inc;
    if (lane == 0) { H.tot2 = tot2; H.gkey = gkey; H.all_ok = 1; }
  }
  __syncthreads();
}
// every rank's header through its self-validating words (all threads call): thread (r, w) polls word w of rank r and keeps its half in
// `ll`; then thread r holds rank r's 12 header words in the order of spec_load_headers.  false: a word did not come within the bound
__device__ __forceinline__ bool peer_headers_ll(SpecHdrRegs& hregs, uint32_t* __restrict__ ll, unsigned int& s_late, const unsigned int* __restrict__ peer_flags,
                                                uint32_t peer_seq, unsigned long long wait_ticks, int world) {
  if (threadIdx.x == 0) s_late = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < (uint32_t)world * CSSM_PEER_LL_WORDS; i += CSSM_BLOCK) {
    const uint32_t r = i / CSSM_PEER_LL_WORDS, w = i % CSSM_PEER_LL_WORDS;
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(peer_flags + (size_t)r * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_LL) + w;
    unsigned int half = 0u;
    if (!peer_poll_ll(f, peer_seq, wait_ticks, half)) { s_late = 1u | (r << 8); break; }   // (bits 8-15: whose word it was -- Scalars::wait_code)
    ll[i] = half;
  }
  __syncthreads();
  if (s_late) return false;
#pragma unroll
  for (int k = 0; k < 12; ++k) hregs.w[k] = 0.0;
  if ((int)threadIdx.x < world) {
    const uint32_t* h = ll + threadIdx.x * CSSM_PEER_LL_WORDS;
    auto hw = [&](int k) { return cssm_u2d((unsigned long long)h[2 * k] | ((unsigned long long)h[2 * k + 1] << 32)); };
    // (the order of spec_load_headers: count, S, base, the two block totals, S2, the key)
    hregs.w[0] = hw(0); hregs.w[1] = hw(1); hregs.w[2] = hw(2); hregs.w[3] = hw(6); hregs.w[4] = hw(7); hregs.w[5] = hw(8); hregs.w[6] = hw(9);
    hregs.w[7] = hw(10); hregs.w[8] = hw(11); hregs.w[9] = hw(3); hregs.w[10] = hw(4); hregs.w[11] = hw(5);
  }
  return true;
}

#define CSSM_PEER_FLAG_HELLO 8    /* word of a source rank's flag pair that cssm_pf_shard_peer_handshake uses */
// One round of the protocol with nothing attached, run by every rank at once right after the windows were mapped: thread q writes a
// token where rank q looks for this rank's (system-scope release) and waits, bounded, for rank q's token in this rank's own flags.
// result[0] = number of ranks whose token did not arrive.  A rank whose mapping, peer access or cross-GPU visibility does not work
// shows up here, on the host, before a series depends on it.
// ... with a PAYLOAD: before the token, thread q stores CSSM_PEER_PROBE_WORDS doubles -- a function of (token, source rank, index) --
// at the head of segment `rank` of rank q's window 0, with the plain stores + system-scope fence + release the pack blocks use for rows;
// k_peer_verify, a launch of its own BEHIND this one, reads what the peers left in this rank's window with the loads the propagate reads
// rows with (system-scope) AND with plain ones, and counts what differs.  The host runs two rounds with different tokens: a line of
// round one that some cache of this GPU kept would answer round two's plain loads with the wrong pattern (reported, not fatal: plain
// loads of a window are not on any data path).
#define CSSM_PEER_PROBE_WORDS 16
__device__ __forceinline__ double peer_probe_word(uint32_t token, int src_rank, int i) {
  return cssm_u2d(0x3ff0000000000000ull | ((unsigned long long)token << 16) | ((unsigned long long)(src_rank & 0xff) << 8) | (unsigned long long)(i & 0xff));
}
__global__ void k_peer_handshake(const PeerTable* __restrict__ peer, int world, int rank, uint32_t token, unsigned int* __restrict__ result,
                                 unsigned long long wait_ticks, size_t seg) {
  __shared__ unsigned int s_missing;
  if (threadIdx.x == 0) s_missing = 0u;
  __syncthreads();
  const int q = (int)threadIdx.x;
  if (q < world) {
    double* w = peer->win[0][q] + (size_t)rank * seg;
    for (int i = 0; i < CSSM_PEER_PROBE_WORDS; ++i) w[i] = peer_probe_word(token, rank, i);
    __threadfence_system();
    __hip_atomic_store(peer->flag[0][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_HELLO, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int* f = peer->flag[0][rank] + (size_t)q * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_HELLO;
    if (!peer_poll_u32(f, token, wait_ticks)) atomicAdd(&s_missing, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) result[0] = s_missing;
}

#ifdef CSSM_OFF_STAMPS
__device__ unsigned long long g_spec_stamps[2048 * 8];   // diagnostic build: clock stamps of the exchange kernels' blocks (tools/archive/exchange_stamps.py)
#define CSSM_SPEC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_spec_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSSM_SPEC_STAMP(k) do { } while (0)
#endif
// result[1] += probe words of this rank's window 0 that system-scope loads read wrongly, result[2] += those that plain loads read wrongly
__global__ void k_peer_verify(const double* __restrict__ win0, int world, uint32_t token, size_t seg, unsigned int* __restrict__ result) {
  const int q = (int)(threadIdx.x / CSSM_PEER_PROBE_WORDS), i = (int)(threadIdx.x % CSSM_PEER_PROBE_WORDS);
  for (int r = q; r < world; r += (int)(blockDim.x / CSSM_PEER_PROBE_WORDS)) {
    const double* w = win0 + (size_t)r * seg + i;
    const unsigned long long want = cssm_d2u(peer_probe_word(token, r, i));
    if (cssm_d2u(ld_sys_f64(w)) != want) atomicAdd(&result[1], 1u);
    if (cssm_d2u(*reinterpret_cast<const volatile double*>(w)) != want) atomicAdd(&result[2], 1u);
  }
}

// grid (tiles of the block + 1 for the header, destination rank); the weights are those k_propagate<SUMS> summed and stored,
// exp(min(w - c, REF_BELOW)) -- or, with the level taken from the global max, exp(w - level) of the stored log-weights
// peer != nullptr: the peer-written exchange -- `out` unused, segment rank -> q lands in peer->win[parity][q] + rank * seg,
// tickets[q] counts the finished blocks of destination q (left at zero again by the block that takes the last ticket)
// Four consecutive rows of a boundary block (one thread's) in 16-byte accesses: the states of particles p0 .. p0 + 3 (struct of arrays: two
// loads per component), the rows (D states + the low word of the row's cumulative weight, 4 (D + 1) doubles in a row) and the four high
// words.  The windows are fine-grained memory: every store is a transaction of its own on the way out, and 8 bytes at a time the 1024
// rows of a tile took 4.5-5 us of the 11 between the launch's start and the eager rows' flag (tools/archive/pack_stamps_local.py).  The caller
// has checked that all three addresses are 16-byte aligned.
template <int D>
__device__ __forceinline__ void pack_rows4(const double* __restrict__ src, const size_t stride, const uint64_t p0, double* __restrict__ orow,
                                           double* __restrict__ ohi, const cssm_u128 (&P)[4]) {
  constexpr int R = D + 1;
  double out[4 * R];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double2* s2 = reinterpret_cast<const double2*>(src + (size_t)k * stride + (size_t)p0);
    const double2 a = s2[0], b = s2[1];
    out[0 * R + k] = a.x; out[1 * R + k] = a.y; out[2 * R + k] = b.x; out[3 * R + k] = b.y;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r * R + D] = cssm_u2d(P[r].lo);
  double2* o2 = reinterpret_cast<double2*>(orow);
#pragma unroll
  for (int j = 0; j < 2 * R; ++j) o2[j] = make_double2(out[2 * j], out[2 * j + 1]);
  double2* h2 = reinterpret_cast<double2*>(ohi);
  h2[0] = make_double2(cssm_u2d(P[0].hi), cssm_u2d(P[1].hi));
  h2[1] = make_double2(cssm_u2d(P[2].hi), cssm_u2d(P[3].hi));
}

template <int RSC = -1>   // the resampler, where the launch knows it at compile time (else PackNeed::rs)
__device__ __forceinline__ void boundary_pack_block(const uint32_t bx, const uint32_t gx, const int q,
                                                              const double* __restrict__ src, size_t stride, const double* __restrict__ logw,
                                                              uint64_t n_local, int d, int world, int rank, long long cap,
                                                              const StepRec* __restrict__ rec, const cssm_u128* __restrict__ subS,
                                                              const cssm_u128* __restrict__ subS2, uint32_t nsub,
                                                              const Scalars* __restrict__ sc, double* __restrict__ out, uint64_t chunk,
                                                              int level_from_max, cssm_u128* __restrict__ pre_out,
                                                              const PeerTable* __restrict__ peer, int parity, uint32_t seq,
                                                              unsigned int* __restrict__ tickets, unsigned int* __restrict__ pre_flag,
                                                              const int grp_set, const PackNeed& xnr, const bool need_on, const int phase, const bool merged = false) {
  // merged: this block runs in the merged launch (k_exchange_offspring), next to the offspring blocks that consume what it reads
  const PackNeed* xn = &xnr;
  if (bx == gx - 2) CSSM_SPEC_STAMP(0);   // (diagnostic build: the header block's first instruction)
  // need_on (peer-written exchange; xn.eager < cap): the eager rows travel at once, of the others those the neighbours need, behind every
  // rank's header (PackNeed).  phase: bit 0 = the header and prefix blocks work, bit 1 = the row blocks write their eager rows, bit 2 = the
  // row blocks write needed rows beyond them -- shards of ONE process that share a stream launch bits 0 | 1 and bit 2 apart, every shard's
  // headers before any shard waits for them: a row block of the first shard would otherwise wait for a header that a launch BEHIND it on
  // the same stream is to write
  // grp_set >= 0: k_propagate's blocks accumulated the sums (and sums of squares) of groups of 32 units in that set of Scalars::grp / grp2:
  // the header block totals 2 x 32 group sums in ONE wave instead of 2 x nsub unit sums in four (it is the head of the exchange's critical
  // path: every offspring block of every rank waits for it -- 3.4 us from entry to flag at 1024 units, tools/archive/exchange_stamps.py)
  // bx / gx / q: the block's place in a (gx, world) grid -- blockIdx.x, gridDim.x, blockIdx.y of k_boundary_pack; the merged
  // exchange + offspring kernel of the peer-written exchange hands its first gx * world blocks through here
  // pre_flag (merged kernel; else nullptr): the prefix block announces pre_out with the exchange number (agent-scope release)
  // pre_out (nsub <= 8 * CSSM_BLOCK, else nullptr): a block of its own (segment 0's last) also writes the EXCLUSIVE prefix of the sub-unit
  // sums -- pre_out[j] = subS[0] + .. + subS[j - 1] -- which k_offspring_expand_spec's block j then reads as one word instead of
  // summing j entries itself (on average 8 KiB per block and a block-wide sum with two barriers)
  // level_from_max: the sums (subS) were formed relative to the level chosen with the GLOBAL max after an all-gather of the
  // local maxima (cssm_pf_shard_sums: LGCP series, whose level is the max; the repetition of a series an outlying observation
  // voided) -- the weights of the rows are then relative to sc->ref and the header carries the global max
  // chunk = particles per sub-unit sum of k_propagate (subS): when the tiles of the carried block coincide with
  // sub-units, the prefix of the tiles before a block's own is read from subS instead of being recomputed
  __shared__ cssm_u128 s_w[CSSM_BLOCK / 64], s_r[2][CSSM_BLOCK / 64];
  __shared__ unsigned int s_need, s_hlate;
  // rows next to the boundary that travel at once (PackNeed; every row where nobody asked for less: the collective exchange, CSSM_PEER_ALL_ROWS)
  const long long eager = (need_on && peer != nullptr) ? xn->eager : cap;
  // The series is on hold (capacity miss), void (level ruled out) or a peer is missing: nothing may change -- IF THE BIT WAS THERE WHEN
  // THIS LAUNCH BEGAN.  The first offspring block of this very launch raises such a bit as soon as it holds every rank's header (its
  // verdict), and a pack block that saw it then returned without its header or its ticket: the rank it was for waited for words that never
  // came.  Reading the bits at the block's entry is not enough -- the blocks of a launch start XCD by XCD, and with three processes on one
  // GPU a header block was seen to start microseconds behind the offspring blocks of its own launch (tools/ipc_soak.py: one run in six at a
  // capacity miss).  So the question is put so that every block of a launch answers it alike: the bits count only if the observation that
  // raised them (Scalars::fail_step, written with them) is not this launch's own.  (fail_step still unset: the bit is being raised right
  // now -- by this launch.)  Tested where a block is about to store: at the head it was a round trip of its own.
  const uint32_t err_now = __hip_atomic_load(&sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (4u | 8u | 16u);
  const uint32_t fail_now = __hip_atomic_load(&sc->fail_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (asked where a block is about to store or to wait: the two loads are on their way from here, the wait for them is there -- asked here,
  //  it stood a round trip in front of every other load of the block: 1.1 us, tools/archive/pack_stamps_local.py)
  auto held = [&]() -> bool { return err_now != 0u && fail_now != 0xffffffffu && fail_now != rec->step; };
  const long long R = d + 1, HD = spec_hdr(d), seg = spec_seg(d, cap);
  double* oseg = peer ? peer->win[parity][q] + (size_t)rank * seg : out + (size_t)q * seg;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // only the two adjacent ranks can own slots of this rank's boundary particles (k_offspring_expand_spec's verdict refuses
  // anything else), so only their segments carry rows; every other segment is its header
  const long long cnt = (q == rank + 1 || q == rank - 1) ? ((long long)n_local < cap ? (long long)n_local : cap) : 0;
  // peer-written exchange: this block's part of segment rank -> q is done (all threads call; see PeerTable)
  auto peer_done = [&](bool header) {
    if (peer == nullptr) return;
    if (header) {   // (thread 0 alone wrote the header: its release orders those stores before the flag)
      if (threadIdx.x == 0)
        __hip_atomic_store(peer->flag[parity][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  };
  // rows this block wrote are done (all threads call).  extra = false: its EAGER rows -- the last block's ticket sets the ROWS flag;
  // extra = true: its needed rows beyond them -- the block's count joins the destination's maximum, the last ticket publishes it (NEED) and
  // sets the EXTRA flag
  auto rows_done = [&](const bool extra) {
    if (peer == nullptr) return;
    // every wave waits for its stores to have left it, the block meets, and ONE wave fences at system scope (a fence is a write-back of
    // the XCD's L2: four waves each issuing their own stood 1.7-2.4 us between the last row and the ticket, tools/archive/pack_stamps_local.py)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __threadfence_system();
    if (!extra) CSSM_SPEC_STAMP(7);
    if (threadIdx.x == 0) {
      unsigned int* f = peer->flag[parity][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE;
      const unsigned int nblk = gx - 2u;                // (the row blocks of this destination)
      if (!extra) {
        const unsigned int t = atomicAdd(&tickets[q], 1u);
        if (t + 1u == nblk) {
          tickets[q] = 0u;                              // (the next launch on this stream starts from zero)
          // (every block fenced in front of its ticket, this one included, and the ticket's return is behind all of them; the flag is a
          //  RELEASE store all the same -- round 5 stored it relaxed and saved this block a second write-back of its XCD's L2 (~0.8 us
          //  on a flag that is long set when its reader gets to it), but the tickets are relaxed device-scope atomics and the order
          //  "every block's rows, then the flag" across a link has never run on two physical GPUs: the memory model's guarantee, not
          //  an argument about when stores are acknowledged, is what a reader on another GPU gets)
          __hip_atomic_store(f + CSSM_PEER_FLAG_ROWS, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          CSSM_SPEC_STAMP(3);   // (diagnostic build: a pack block that set a destination's ROWS flag -- tools/archive/pack_stamps_local.py)
        }
      } else {
        unsigned int* tk = tickets + CSSM_PEER_TICKET_EXTRA;
        const unsigned int mine = s_need;
        if (mine) { atomicMax(&xn->need[q], mine); __threadfence(); }
        const unsigned int t = atomicAdd(&tk[q], 1u);
        if (t + 1u == nblk) {
          tk[q] = 0u;
          __threadfence();
          const unsigned int nn = atomicExch(&xn->need[q], 0u);
          if (cnt > 0) {
            const unsigned int eg = (unsigned int)((long long)cnt < eager ? (long long)cnt : eager);
            atomicAdd(&xn->stat[0], (unsigned long long)(nn > eg ? nn : eg)); atomicAdd(&xn->stat[1], 1ull);
            if (nn > eg) atomicAdd(&xn->stat[2], 1ull);
          }
          __hip_atomic_store(f + CSSM_PEER_FLAG_NEED, nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __builtin_amdgcn_s_waitcnt(0);                // (the count has arrived before the flag leaves; the rows: fenced in front of the tickets)
          __hip_atomic_store(f + CSSM_PEER_FLAG_EXTRA, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (release: as the ROWS flag)
        }
      }
    }
  };
  const uint64_t first = (q < rank) ? 0 : n_local - (uint64_t)cnt;     // first particle of the block the segment carries
  const double cref = level_from_max ? sc->ref : rec->ref;
  auto tile_weights = [&](uint64_t base, cssm_u128 (&qq)[CSSM_ITEMS]) {   // particles first + base + 4 tid .. of the block
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      const uint64_t i = base + (uint64_t)threadIdx.x * CSSM_ITEMS + r;
      // level_from_max: log-weights, rescaled as k_tile_sums does; else the weights k_propagate<SUMS> stored in their place
      qq[r] = (i < (uint64_t)cnt) ? cssm_fix_from_unit(level_from_max ? cssm_exp_le0(logw[first + i] - cref) : logw[first + i]) : cssm_u128_zero();
    }
  };
  auto block_total = [&](cssm_u128 v) -> cssm_u128 {   // sum over the block's threads (uniform result)
    v = wave_sum_u128(v);
    __syncthreads();
    if (lane == 0) s_w[wid] = v;
    __syncthreads();
    cssm_u128 t = s_w[0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) t = cssm_u128_add(t, s_w[w]);
    return t;
  };
  // grid.x = tiles of the block + 2: the header has a block of its own, and so have the prefixes of the sub-unit sums (pre_out; in the
  // header block they lengthened the launch's longest latency chain by 1.6 us)
  if (bx == gx - 1) {   // the prefix block: thread t owns the E consecutive entries from t E on (E = 4; 8 beyond 1024 sub-unit sums)
    if (pre_out == nullptr || q != 0 || !(phase & 1)) return;
    // (the entries are requested ahead of the hold test)
    __shared__ cssm_u128 s_p[CSSM_BLOCK / 64];
    constexpr int EMAX = 8;
    const uint32_t E = (nsub > 4u * CSSM_BLOCK) ? (uint32_t)EMAX : 4u;   // (uniform)
    cssm_u128 v[EMAX];
#pragma unroll
    for (int k = 0; k < EMAX; ++k) { const uint32_t i = threadIdx.x * E + (uint32_t)k; v[k] = ((uint32_t)k < E && i < nsub) ? subS[i] : cssm_u128_zero(); }
    cssm_u128 run[EMAX];   // run[k] = v[0] + .. + v[k - 1]
    cssm_u128 own = cssm_u128_zero();
#pragma unroll
    for (int k = 0; k < EMAX; ++k) { run[k] = own; own = cssm_u128_add(own, v[k]); }
    const cssm_u128 inc = wave_scan_u128(own, lane);
    if (lane == 63) s_p[wid] = inc;
    __syncthreads();
    cssm_u128 ex = inc;                                  // exclusive prefix of the thread's first entry: inc - own + the waves before
    ex.hi = inc.hi - own.hi - (inc.lo < own.lo ? 1ull : 0ull); ex.lo = inc.lo - own.lo;
    for (int w = 0; w < wid; ++w) ex = cssm_u128_add(ex, s_p[w]);
    const uint32_t i0 = threadIdx.x * E;
    if (held()) return;
#pragma unroll
    for (int k = 0; k < EMAX; ++k) if ((uint32_t)k < E && i0 + (uint32_t)k < nsub) pre_out[i0 + (uint32_t)k] = cssm_u128_add(ex, run[k]);
    if (pre_flag != nullptr) {   // (merged kernel: the offspring blocks of this very launch read them, behind this flag)
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(pre_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    CSSM_SPEC_STAMP(6);
    return;
  }
  const bool header_block = (bx == gx - 2);
  if (header_block && !(phase & 1)) return;
  // every block of every rank's launch waits for what this block writes: its waves go first where they share a CU with others (header
  // words seen by the offspring blocks 3.9 instead of 4.1 us into the merged launch, same-box A/B: tools/archive/ab_header_prio.sh)
  if (header_block) __builtin_amdgcn_s_setprio(3);
  if (header_block) CSSM_SPEC_STAMP(4);
  if (!header_block) {
  // stage A (phase bit 1): the eager rows of this tile; stage B (bit 2): its needed rows beyond them, behind every rank's header
  const bool do_a = (phase & 2) != 0, do_b = (phase & 4) != 0;
  if (!do_a && !do_b) return;
  if (threadIdx.x == 0) s_need = 0u;
  if (cnt == 0) return;   // (not a neighbour: no rows, and nobody waits for a rows flag of this segment)
  CSSM_SPEC_STAMP(0);
  // this tile's rows [base, tile_end) of the block; the eager rows are [e_lo, e_hi): the block's last E for rank + 1, its first E for rank - 1
  const uint64_t base = (uint64_t)bx * CSSM_TILE;
  const long long E = (eager < cnt) ? (eager > 0 ? eager : 0) : cnt;
  const long long e_lo = (q > rank) ? cnt - E : 0, e_hi = (q > rank) ? cnt : E;
  const long long tile_end = ((long long)base + CSSM_TILE < cnt) ? (long long)base + CSSM_TILE : cnt;
  const bool work_a = do_a && e_lo < tile_end && (long long)base < e_hi;                                // (uniform)
  const bool work_b = do_b && E < cnt && !(e_lo <= (long long)base && tile_end <= e_hi);                // (uniform: rows outside the eager range)
  bool have_headers = false;
  auto wait_headers = [&]() -> bool {   // every rank's header -> *xn->H (the offspring blocks of this launch wait for the same words)
    SpecHdrRegs hregs;
    if (!peer_headers_ll(hregs, xn->ll, s_hlate, xn->my_flags, seq, sc->peer_wait_ticks, world)) {
      if (threadIdx.x == 0) {
        Scalars* scw = const_cast<Scalars*>(sc);
        atomicOr(&scw->err, 16u); atomicMin(&scw->fail_step, rec->step); atomicOr(&scw->wait_code, 4u | (s_hlate & 0xff00u));
      }
      return false;   // (a peer's header did not come: no ticket, no flag -- the series ends on every rank like one on hold)
    }
    spec_store_headers(*xn->H, hregs, world, cap);
    have_headers = true;
    return true;
  };
  // (a tile without eager rows: the headers ahead of the tile's own loads, so that the header words and the tile's sums are not held in
  //  registers together)
  if (work_b && !work_a) { if (held()) return; if (!wait_headers()) return; }
  cssm_u128 run0 = cssm_u128_zero(), tsum = cssm_u128_zero();   // exclusive prefix of the thread's first row inside the block; sum of its rows
  // the tile's own weights are requested FIRST (raw: log-weights or stored weights), ahead of the loads and barriers of the prefix of the
  // tiles before it -- one round trip instead of two on the way to the eager rows' flag.  (Stage A writing its rows from these instead
  // of reading them once more: 95 -> 97 VGPR in the merged kernel, a wave of occupancy.)
  double wraw[CSSM_ITEMS] = {0.0, 0.0, 0.0, 0.0};
  if (work_a || work_b) {
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      const uint64_t i = base + (uint64_t)threadIdx.x * CSSM_ITEMS + r;
      if (i < (uint64_t)cnt) wraw[r] = logw[first + i];
    }
  }
  if (work_a || work_b) {
  // prefix of the tiles before this block's tile
  cssm_u128 toff = cssm_u128_zero();
  // The carried block starts on a boundary of the sub-units whose sums k_propagate (or k_tile_sums) formed -- chunk particles
  // each, a whole number of tiles: the sub-units wholly before this tile come from subS, at most chunk / 1024 - 1 tiles of the
  // tile's own sub-unit are formed again.  (Round 2 recognised alignment only for chunk = one tile and otherwise formed EVERY
  // tile before its own again, in every block: quadratic in the capacity -- 29 us per observation for an LGCP shard of 2^21.)
  const bool aligned = (chunk % (uint64_t)CSSM_TILE == 0) && (first % chunk == 0);
  uint32_t t_begin = 0;
  if (aligned) {
    const uint32_t per = (uint32_t)(chunk / CSSM_TILE);               // tiles per sub-unit
    const uint32_t c0 = (uint32_t)(first / chunk), nc = bx / per;   // sub-units wholly before this tile
    // (the kernel is one latency chain after another at the sizes it runs at -- a capacity of a few thousand rows: what can be
    //  requested together is: up to 8 sums in flight, the rest in a loop)
    if (nc <= 8u) {   // (uniform)
      cssm_u128 pre8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) pre8[c] = ((uint32_t)c < nc) ? subS[c0 + (uint32_t)c] : cssm_u128_zero();
#pragma unroll
      for (int c = 0; c < 8; ++c) toff = cssm_u128_add(toff, pre8[c]);
    } else {
      // more of them -- the LAST tiles of a boundary block of thousands of rows, i.e. the eager rows for rank + 1, whose flag the
      // neighbour's expansion blocks wait for: one sum per thread and a block total instead of a loop of dependent loads
      cssm_u128 v = cssm_u128_zero();
      for (uint32_t c = threadIdx.x; c < nc; c += CSSM_BLOCK) v = cssm_u128_add(v, subS[c0 + c]);
      toff = block_total(v);
    }
    t_begin = nc * per;
  }
  for (uint32_t t = t_begin; t < bx; ++t) {
    cssm_u128 qq[CSSM_ITEMS];
    tile_weights((uint64_t)t * CSSM_TILE, qq);
    cssm_u128 a = cssm_u128_zero();
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) a = cssm_u128_add(a, qq[r]);
    toff = cssm_u128_add(toff, block_total(a));
  }
  // this tile: the threads' exclusive prefixes
#pragma unroll
  for (int r = 0; r < CSSM_ITEMS; ++r) {
    const uint64_t i = base + (uint64_t)threadIdx.x * CSSM_ITEMS + r;
    if (i < (uint64_t)cnt) tsum = cssm_u128_add(tsum, cssm_fix_from_unit(level_from_max ? cssm_exp_le0(wraw[r] - cref) : wraw[r]));
  }
  const cssm_u128 inc = wave_scan_u128(tsum, lane);
  __syncthreads();
  if (lane == 63) s_w[wid] = inc;
  __syncthreads();
  cssm_u128 run = toff;
  for (int w = 0; w < wid; ++w) run = cssm_u128_add(run, s_w[w]);
  run = cssm_u128_add(run, inc);
  run0.lo = run.lo - tsum.lo; run0.hi = run.hi - tsum.hi - (run.lo < tsum.lo ? 1u : 0u);
  }
  // rows [lo, hi) of the thread's four -> the segment (the weights are read once more, through a pointer the compiler cannot identify with
  // the first one: four 128-bit weights held across the header wait and the counts cost the merged kernel a wave of occupancy)
  const uint64_t i0 = base + (uint64_t)threadIdx.x * CSSM_ITEMS;
  auto write_rows = [&](const long long lo, const long long hi, const bool pstart) {
    const double* lw = logw;
    asm volatile("" : "+v"(lw));
    cssm_u128 run = run0;
    static_assert(CSSM_ITEMS == 4, "pack_rows4: a thread's four rows");
    if ((long long)i0 >= lo && (long long)i0 + CSSM_ITEMS <= hi && (long long)i0 + CSSM_ITEMS <= cnt && d >= 1 && d <= 4) {
      // all four rows travel: 16-byte loads and stores where the addresses allow (else the row-by-row path below)
      double* orow = oseg + HD + (long long)i0 * R;
      double* ohi = oseg + HD + cap * R + (long long)i0;
      const uint64_t p0 = first + i0;
      if ((((uintptr_t)orow | (uintptr_t)ohi | (uintptr_t)(src + p0) | (uintptr_t)(logw + p0) | (uintptr_t)(stride * sizeof(double))) & 15u) == 0u) {
        if (pstart && (long long)i0 == lo && i0 > 0) {
          unsigned long long* ps = reinterpret_cast<unsigned long long*>(peer->flag[parity][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_PSTART);
          ps[0] = run.lo; ps[1] = run.hi;
        }
        const double2* w2 = reinterpret_cast<const double2*>(lw + p0);
        const double2 wa = w2[0], wb = w2[1];
        const double wv[4] = {wa.x, wa.y, wb.x, wb.y};
        cssm_u128 P[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { run = cssm_u128_add(run, cssm_fix_from_unit(level_from_max ? cssm_exp_le0(wv[r] - cref) : wv[r])); P[r] = run; }
        if (d == 3) pack_rows4<3>(src, stride, p0, orow, ohi, P);
        else if (d == 1) pack_rows4<1>(src, stride, p0, orow, ohi, P);
        else if (d == 2) pack_rows4<2>(src, stride, p0, orow, ohi, P);
        else pack_rows4<4>(src, stride, p0, orow, ohi, P);
        return;
      }
    }
#pragma unroll
    for (int r = 0; r < CSSM_ITEMS; ++r) {
      const long long i = (long long)i0 + r;
      if (i < cnt) {
        if (pstart && i == lo && i > 0) {   // the cumulative weight in front of the first eager row (the reader's start slot of that row)
          unsigned long long* ps = reinterpret_cast<unsigned long long*>(peer->flag[parity][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_PSTART);
          ps[0] = run.lo; ps[1] = run.hi;
        }
        run = cssm_u128_add(run, cssm_fix_from_unit(level_from_max ? cssm_exp_le0(lw[first + (uint64_t)i] - cref) : lw[first + (uint64_t)i]));
        if (i >= lo && i < hi) {
          double* o = oseg + HD + i * R;
          for (int k = 0; k < d; ++k) o[k] = src[(size_t)k * stride + (size_t)(first + (uint64_t)i)];
          o[d] = cssm_u2d(run.lo);
          oseg[HD + cap * R + i] = cssm_u2d(run.hi);
        }
      }
    }
  };
  if (held()) return;   // (on hold when the launch began: no rows, no tickets -- every block of the launch alike)
  if (work_a) __builtin_amdgcn_s_setprio(2);   // (the neighbour's expansion blocks wait for the eager rows' flag)
  if (do_a) {
    if (work_a) CSSM_SPEC_STAMP(1);
    if (work_a && (long long)i0 < e_hi && (long long)i0 + CSSM_ITEMS > e_lo) write_rows(e_lo, e_hi, peer != nullptr && q > rank);
    if (work_a) CSSM_SPEC_STAMP(2);
    rows_done(false);
  }
  if (do_b) {
    if (work_b) {
      if (!have_headers && !wait_headers()) return;
      // the slot counts of this thread's rows as the READER forms them in expand_spec_body -- the same header words, the same arithmetic,
      // the same numbers
      SpecHeaders& H = *xn->H;
      cssm_u128 off = H.off[rank];
      if (q > rank) {
        const cssm_u128 Sr = H.S[rank], Pr = H.phigh[rank];
        cssm_u128 bs; bs.lo = Sr.lo - Pr.lo; bs.hi = Sr.hi - Pr.hi - (Sr.lo < Pr.lo ? 1u : 0u);
        off = cssm_u128_add(off, bs);
      }
      const double totd = cssm_u128_to_double(H.tot), u = rec->u;
      const uint64_t n_global = xn->n_global;
      const bool pow2 = (n_global & (n_global - 1)) == 0;
      const double inv_n = 1.0 / (double)n_global;
      const uint32_t rstep = rec->step;
      const int rs = (RSC >= 0) ? RSC : xn->rs;
      auto count_of = [&](cssm_u128 G) -> uint64_t {
        if (cssm_u128_is_zero(G)) return 0;
        const double C = cssm_u128_to_double(G) / totd;
        if (rs == CSSM_RESAMPLE_STRATIFIED) return cssm_strat_count(C, xn->seed, rstep, n_global);
        return pow2 ? cssm_sys_count_pow2(C, u, n_global, inv_n) : cssm_sys_count(C, u, n_global);
      };
      // ONE count per thread: its four rows travel together.  Block for rank + 1 (a suffix is needed: end slots grow with the row): the
      // thread's rows travel iff the LAST of them ends beyond this rank's last slot; block for rank - 1 (a prefix): iff the FIRST of them
      // starts below this rank's first slot.  Up to three rows more than needed at the edge; the reader clamps their runs to nothing.
      const uint64_t c = count_of(cssm_u128_add(off, (q < rank) ? run0 : cssm_u128_add(run0, tsum)));
      unsigned int mine = 0u;
      bool need_rows = false;
      if (i0 < (uint64_t)cnt && ((q < rank) ? (c < xn->slot_lo) : (c > xn->slot_hi))) {
        need_rows = true;
        const uint64_t i1 = (i0 + CSSM_ITEMS < (uint64_t)cnt) ? i0 + CSSM_ITEMS : (uint64_t)cnt;   // one past the thread's last row
        mine = (q < rank) ? (unsigned int)i1 : (unsigned int)((uint64_t)cnt - i0);
      }
      mine = (unsigned int)wave_max_u64((unsigned long long)mine);
      if (lane == 0 && mine) atomicMax(&s_need, mine);   // (rows_done's barrier stands between this and its reader)
      // (its rows outside the eager range: the eager ones went in stage A)
      if (need_rows) { if (q > rank) write_rows(0, e_lo, false); else write_rows(e_hi, cnt, false); }
    }
    rows_done(true);
  }
  return;
  }
  // header: the rank's totals of the sub-unit sums k_propagate formed, the key of its max, base.  Everything it reads is
  // requested before anything is summed: up to 4 sums and 4 sums of squares per thread (1024 sub-units), the max slots, the
  // boundary blocks' sub-unit sums (below).
  unsigned long long key = 0ull;
  const long long cnt_all = ((long long)n_local < cap) ? (long long)n_local : cap;
  cssm_u128 ptot[2];
  {
    // group sums at hand and both boundary blocks made of whole units: one round of loads, one wave each -- wave 0 the totals (lanes 0-31
    // the groups' sums, lanes 32-63 their sums of squares), wave 1 the key of the max, waves 2 and 3 the boundary blocks' totals
    const bool whole = (chunk % (uint64_t)CSSM_TILE == 0) && ((uint64_t)cnt_all % chunk == 0) && ((n_local - (uint64_t)cnt_all) % chunk == 0);
    if (grp_set >= 0 && whole && !level_from_max) {   // (uniform)
      static_assert(CSSM_GRP_SMALL == 32, "one wave holds the groups' sums and their sums of squares (a shard: layout 1, at most 32 groups)");
      __shared__ cssm_u128 s_pt[2];
      __shared__ unsigned long long s_hkey;
      if (wid == 0) {
        const size_t at = ((size_t)grp_set * 2 * CSSM_GRP_MAX + (size_t)(lane & 31)) * CSSM_SLOT_STRIDE;
        const unsigned long long* g = (lane < 32) ? &sc->grp[at] : &sc->grp2[at];
        const unsigned long long l0 = g[0], l1 = g[(size_t)CSSM_GRP_MAX * CSSM_SLOT_STRIDE];   // the two limb sums a0, a1 (< 2^61 each) -> a0 + a1 2^56
        cssm_u128 x, y;
        x.lo = l0; x.hi = 0ull;
        y.lo = l1 << CSSM_GRP_LIMB; y.hi = l1 >> (64 - CSSM_GRP_LIMB);
        const cssm_u128 inc = wave_scan_u128(cssm_u128_add(x, y), lane);
        cssm_u128 S, T;
        S.lo = readlane_u64(inc.lo, 31); S.hi = readlane_u64(inc.hi, 31);
        T.lo = readlane_u64(inc.lo, 63); T.hi = readlane_u64(inc.hi, 63);
        if (lane == 0) {
          cssm_u128 S2; S2.lo = T.lo - S.lo; S2.hi = T.hi - S.hi - (T.lo < S.lo ? 1ull : 0ull);
          s_r[0][0] = S; s_r[1][0] = S2;
#pragma unroll
          for (int w = 1; w < CSSM_BLOCK / 64; ++w) { s_r[0][w] = cssm_u128_zero(); s_r[1][w] = cssm_u128_zero(); }
        }
      } else if (wid == 1) {
        unsigned long long k = (lane < CSSM_MAXSLOTS) ? sc->maxslot[(size_t)lane * CSSM_SLOT_STRIDE] : 0ull;   // slot set 0 (sharded handles)
        k = wave_max_u64(k);
        if (lane == 0) s_hkey = k;
      } else {
        const int which = wid - 2;
        const uint64_t bfirst = which ? n_local - (uint64_t)cnt_all : 0;
        const uint32_t b0 = (uint32_t)(bfirst / chunk), nch = (uint32_t)((uint64_t)cnt_all / chunk);
        cssm_u128 c = cssm_u128_zero();
        for (uint32_t t = (uint32_t)lane; t < nch; t += 64u) c = cssm_u128_add(c, subS[b0 + t]);
        c = wave_sum_u128(c);
        if (lane == 0) s_pt[which] = c;
      }
      __syncthreads();
      CSSM_SPEC_STAMP(1);
      key = s_hkey;
      ptot[0].lo = s_pt[0].lo; ptot[0].hi = s_pt[0].hi; ptot[1].lo = s_pt[1].lo; ptot[1].hi = s_pt[1].hi;
    } else {
  if (level_from_max) {
    key = cssm_order_key(sc->gmax);                    // (the slots were exported and cleared before the all-gather)
  } else if (threadIdx.x < 64) {
    key = (threadIdx.x < CSSM_MAXSLOTS) ? sc->maxslot[(size_t)threadIdx.x * CSSM_SLOT_STRIDE] : 0ull;   // slot set 0 (sharded handles)
  }
  cssm_u128 a4[4], b4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = threadIdx.x + (uint32_t)k * CSSM_BLOCK;
    a4[k] = (i < nsub) ? subS[i] : cssm_u128_zero();
    b4[k] = (i < nsub) ? subS2[i] : cssm_u128_zero();
  }
  cssm_u128 p2[2] = {cssm_u128_zero(), cssm_u128_zero()};
  bool al2[2];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const uint64_t bfirst = which ? n_local - (uint64_t)cnt_all : 0;
    al2[which] = (chunk % (uint64_t)CSSM_TILE == 0) && (bfirst % chunk == 0) && ((uint64_t)cnt_all % chunk == 0);
    if (al2[which]) {
      const uint32_t b0 = (uint32_t)(bfirst / chunk), nch = (uint32_t)((uint64_t)cnt_all / chunk);
      if (threadIdx.x < nch) p2[which] = subS[b0 + threadIdx.x];            // (the first 256 of them; a loop for more below)
    }
  }
  cssm_u128 a = cssm_u128_zero(), b = cssm_u128_zero();
#pragma unroll
  for (int k = 0; k < 4; ++k) { a = cssm_u128_add(a, a4[k]); b = cssm_u128_add(b, b4[k]); }
  for (uint32_t i = threadIdx.x + 4u * CSSM_BLOCK; i < nsub; i += CSSM_BLOCK) { a = cssm_u128_add(a, subS[i]); b = cssm_u128_add(b, subS2[i]); }
  a = wave_sum_u128(a); b = wave_sum_u128(b);
  __syncthreads();
  if (lane == 0) { s_r[0][wid] = a; s_r[1][wid] = b; }
  __syncthreads();
  if (!level_from_max && threadIdx.x < 64) key = wave_max_u64(key);
  // total weights of the rank's FIRST-cap and LAST-cap blocks (both travel in every header; the LAST one gives the base)
  const uint32_t ntile = (uint32_t)((cnt_all + CSSM_TILE - 1) / CSSM_TILE);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const uint64_t bfirst = which ? n_local - (uint64_t)cnt_all : 0;
    const bool al = al2[which];
    cssm_u128 acc = cssm_u128_zero();
    if (al) {   // whole sub-units: their sums are k_propagate's (k_tile_sums')
      const uint32_t b0 = (uint32_t)(bfirst / chunk), nch = (uint32_t)((uint64_t)cnt_all / chunk);
      cssm_u128 c = p2[which];
      for (uint32_t t = threadIdx.x + CSSM_BLOCK; t < nch; t += CSSM_BLOCK) c = cssm_u128_add(c, subS[b0 + t]);
      acc = block_total(c);
    } else {
      for (uint32_t t = 0; t < ntile; ++t) {
        cssm_u128 c = cssm_u128_zero();
#pragma unroll
        for (int r = 0; r < CSSM_ITEMS; ++r) {
          const uint64_t i = (uint64_t)t * CSSM_TILE + (uint64_t)threadIdx.x * CSSM_ITEMS + r;
          if (i < (uint64_t)cnt_all)
            c = cssm_u128_add(c, cssm_fix_from_unit(level_from_max ? cssm_exp_le0(logw[bfirst + i] - cref) : logw[bfirst + i]));
        }
        acc = cssm_u128_add(acc, block_total(c));
      }
    }
    ptot[which] = acc;
  }
    }
  }
  if (held()) return;
  __shared__ unsigned long long s_hw[12];
  if (threadIdx.x == 0) {
    cssm_u128 S = s_r[0][0], S2 = s_r[1][0];
#pragma unroll
    for (int w = 1; w < CSSM_BLOCK / 64; ++w) { S = cssm_u128_add(S, s_r[0][w]); S2 = cssm_u128_add(S2, s_r[1][w]); }
    cssm_u128 bs = cssm_u128_zero();
    if (q > rank) { bs.lo = S.lo - ptot[1].lo; bs.hi = S.hi - ptot[1].hi - (S.lo < ptot[1].lo ? 1u : 0u); }
    s_hw[0] = cssm_d2u((double)cnt);
    s_hw[1] = S.lo; s_hw[2] = S.hi; s_hw[3] = S2.lo; s_hw[4] = S2.hi;
    s_hw[5] = key;
    s_hw[6] = bs.lo; s_hw[7] = bs.hi;
    s_hw[8] = ptot[0].lo; s_hw[9] = ptot[0].hi;
    s_hw[10] = ptot[1].lo; s_hw[11] = ptot[1].hi;
  }
  __syncthreads();
  if (peer != nullptr && threadIdx.x < CSSM_PEER_LL_WORDS) {   // the self-validating copy first: it is what the group-sum readers wait for
    const unsigned long long hw = s_hw[threadIdx.x >> 1];
    const unsigned long long half = (threadIdx.x & 1u) ? (hw >> 32) : (hw & 0xffffffffull);
    unsigned long long* ll = reinterpret_cast<unsigned long long*>(peer->flag[parity][q] + (size_t)rank * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_LL);
    __hip_atomic_store(ll + threadIdx.x, ((unsigned long long)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  CSSM_SPEC_STAMP(2);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < 12; ++w) oseg[w] = cssm_u2d(s_hw[w]);
  }
  peer_done(true);
  // (merged launch: the first offspring block clears the max slots this block has read -- behind the count of the header blocks that are
  //  done with them: CSSM_PEER_TICKET_HDR_DONE)
  if (merged && tickets != nullptr && threadIdx.x == 0) atomicAdd(&tickets[CSSM_PEER_TICKET_HDR_DONE], 1u);
  CSSM_SPEC_STAMP(5);
}

__global__ __launch_bounds__(CSSM_BLOCK) void k_boundary_pack(const double* __restrict__ src, size_t stride, const double* __restrict__ logw,
                                                              uint64_t n_local, int d, int world, int rank, long long cap,
                                                              const StepRec* __restrict__ rec, const cssm_u128* __restrict__ subS,
                                                              const cssm_u128* __restrict__ subS2, uint32_t nsub,
                                                              const Scalars* __restrict__ sc, double* __restrict__ out, uint64_t chunk,
                                                              int level_from_max, cssm_u128* __restrict__ pre_out,
                                                              const PeerTable* __restrict__ peer = nullptr, int parity = 0, uint32_t seq = 0u,
                                                              unsigned int* __restrict__ tickets = nullptr, int grp_set = -1,
                                                              int phase = 7, long long eager = 0, uint64_t n_global = 0, uint64_t seed = 0, int rs = 0,
                                                              uint64_t slot_lo = 0, uint64_t slot_hi = 0) {
  // eager < cap (peer != nullptr): the eager rows travel at once (phase bit 1), of the others those the neighbours need, behind all headers
  // (phase bit 2: PackNeed)
  __shared__ SpecHeaders H;
  __shared__ uint32_t s_ll[64 * CSSM_PEER_LL_WORDS];
  PackNeed xn;
  xn.H = &H; xn.ll = s_ll; xn.my_flags = (peer != nullptr) ? peer->flag[parity][rank] : nullptr;
  xn.need = (tickets != nullptr) ? tickets + CSSM_PEER_TICKET_NEED : nullptr;
  xn.stat = (tickets != nullptr) ? reinterpret_cast<unsigned long long*>(tickets + CSSM_PEER_TICKET_STAT) : nullptr;
  xn.n_global = n_global; xn.seed = seed; xn.slot_lo = slot_lo; xn.slot_hi = slot_hi; xn.rs = rs; xn.eager = eager;
  boundary_pack_block(blockIdx.x, gridDim.x, (int)blockIdx.y, src, stride, logw, n_local, d, world, rank, cap, rec, subS, subS2, nsub, sc, out, chunk,
                      level_from_max, pre_out, peer, parity, seq, tickets, nullptr, grp_set, xn, peer != nullptr && eager < cap, phase);
}

// After the all-to-all: every segment's rows -> the slots of this rank they own.  Global cumulative weight of row i of
// segment s = S_off(s) + base(s) + P_i; end slot = cnt of the contract (the sender's k_offspring arrives at the same
// number for the same particle: its fast path equals the contract's count by construction).  Ancestor index of a slot
// = n_split + row number in the receive buffer (in R-double rows), which k_propagate resolves in place.
// Also: slots of this rank that neither its own particles nor the received rows own -> err bit 3 (8): exact exchange.
// What every block of the launch does first: the headers of all segments -> per-rank sums, offsets, block totals, and the
// verdict "every rank's slots are covered by its own particles plus its neighbours' boundary blocks".  The verdict is a
// function of the headers alone, and every rank holds every header: all ranks arrive at the same verdict without talking.
// stage 2: the verdict (all threads call)
// rs / seed / step: the resampler whose grid the slots follow -- systematic (one uniform u, model/Resampling.scala:63-72) or stratified
// (one uniform per slot from the Philox streams of (seed, step), :78-86: the grid points are keyed by GLOBAL slot, so every rank counts
// the same slots below a cumulative weight)
__device__ __forceinline__ bool spec_read_headers(SpecHeaders& H, int world, int rank, long long cap, int d,
                                                  uint64_t n_local, uint64_t n_global, const double u, int rs = CSSM_RESAMPLE_SYSTEMATIC,
                                                  uint64_t seed = 0, uint32_t step = 0) {
  const double totd = cssm_u128_to_double(H.tot);
  const bool pow2 = (n_global & (n_global - 1)) == 0;
  const double inv_n = 1.0 / (double)n_global;
  auto count_of = [&](cssm_u128 G) -> uint64_t {
    if (cssm_u128_is_zero(G)) return 0;   // (the globally first particle starts at slot 0, as in k_offspring)
    const double C = cssm_u128_to_double(G) / totd;
    if (rs == CSSM_RESAMPLE_STRATIFIED) return cssm_strat_count(C, seed, step, n_global);
    return pow2 ? cssm_sys_count_pow2(C, u, n_global, inv_n) : cssm_sys_count(C, u, n_global);
  };
  // four slot counts per rank (own begin / end, reach of the lower neighbour's last block / of the upper neighbour's first
  // block), one thread each: the verdict is on every block's critical path
  if ((int)threadIdx.x < 4 * world) {
    // (ONE call of count_of for the four kinds: the exact count is a chain of dependent fp64 operations -- a division among them -- and the
    //  four arms of a switch ran one after the other in the block's first wave: 4.3 us from headers to verdict at world 2, on the path of
    //  the launch's last block, tools/archive/exchange_stamps_local.py; the arms now only choose the cumulative weight)
    const int r = (int)(threadIdx.x >> 2), which = (int)(threadIdx.x & 3);
    const int rr = (which == 2) ? r - 1 : ((which == 3) ? r + 1 : r);
    const bool valid = rr >= 0 && rr < world;
    cssm_u128 G = cssm_u128_zero();
    if (valid) {
      const cssm_u128 Sr = H.S[rr];
      cssm_u128 add = cssm_u128_zero();                 // which 0: the rank's first cumulative weight
      if (which == 1) add = Sr;                         // its last
      else if (which == 2) { const cssm_u128 Pp = H.phigh[rr]; add.lo = Sr.lo - Pp.lo; add.hi = Sr.hi - Pp.hi - (Sr.lo < Pp.lo ? 1u : 0u); }   // in front of the lower neighbour's last block
      else if (which == 3) add = H.plow[rr];            // behind the upper neighbour's first block
      G = cssm_u128_add(H.off[rr], add);
    }
    const uint64_t v = count_of(G);
    H.cnts[r][which] = valid ? v : 0;
  }
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const int r = (int)threadIdx.x;
    const uint64_t n_per = (n_global + (uint64_t)world - 1) / (uint64_t)world;
    uint64_t lo = (uint64_t)r * n_per; lo = (lo < n_global) ? lo : n_global;
    uint64_t hi = lo + n_per; hi = (hi < n_global) ? hi : n_global;
    bool ok = true;
    if (lo < hi) {
      const uint64_t own_begin = H.cnts[r][0], own_end = H.cnts[r][1];
      if (lo < own_begin)      // the last-cap block of rank r - 1 must reach down to lo (its end is own_begin by construction)
        ok = (r > 0) && H.cnts[r][2] <= lo && !cssm_u128_is_zero(H.phigh[r - 1]);
      if (own_end < hi)        // the first-cap block of rank r + 1 must reach up to hi (r = world - 1 cannot get here:
        ok = ok && (r + 1 < world) && H.cnts[r][3] >= hi;   //  the last cumulative weight is exactly 1)
    }
    if (!ok) H.all_ok = 0;
  }
  __syncthreads();
  (void)rank; (void)n_local;
  return H.all_ok != 0;
}

// What the first offspring block leaves for the next observation (all its threads call, behind its own particles): max-slot set 0 cleared,
// and the two sets of group sums this observation did not use (grp_cur: the set of this one; the handle's exchanges rotate through the
// three -- the next propagate adds to set grp_cur + 1, the one after to the set the observation before this one used)
// hdr_done / n_hdr (the merged launch; else nullptr): the max slots are read by the launch's OWN header blocks, one per destination rank --
// they lead the grid, but the blocks of a launch start XCD by XCD, and with several processes on one GPU a header block was seen to run
// behind the first offspring block's end: it read cleared slots, its destination got a header with another max than everybody else's, and
// the ranks' levels (LGCP: predicted from the max) and likelihoods drifted apart (tools/ipc_soak.py ... lgcp, world 3).  So the clear waits,
// bounded, for the count of header blocks that are done (usually long reached), and takes the count back to zero.
__device__ __forceinline__ void spec_clear_sets(Scalars* __restrict__ sc, const int grp_cur, unsigned int* __restrict__ hdr_done = nullptr, const unsigned int n_hdr = 0u) {
  if (hdr_done != nullptr) {
    if (threadIdx.x == 0) {
      const unsigned long long ticks = sc->peer_wait_ticks;
      unsigned long long t0 = 0ull;
      unsigned int polls = 0u;
      while (__hip_atomic_load(hdr_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_hdr) {
        if ((++polls & 63u) == 0u) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (t0 == 0ull) t0 = now; else if (now - t0 > ticks) break;   // (a header block that never ran: the launch is beyond help; do not hang)
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __hip_atomic_store(hdr_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (threadIdx.x < CSSM_MAXSLOTS) sc->maxslot[(size_t)threadIdx.x * CSSM_SLOT_STRIDE] = 0ull;
  static_assert(2 * 2 * CSSM_GRP_MAX <= CSSM_BLOCK, "one store per thread and array");
  if (threadIdx.x < 2 * 2 * CSSM_GRP_MAX) {
    const uint32_t tq2 = threadIdx.x;
    const size_t at = ((size_t)((grp_cur + 1 + (int)(tq2 / (2 * CSSM_GRP_MAX))) % CSSM_MAXSETS) * 2 * CSSM_GRP_MAX + tq2 % (2 * CSSM_GRP_MAX)) * CSSM_SLOT_STRIDE;
    sc->grp[at] = 0ull; sc->grp2[at] = 0ull;
  }
}
// The received rows are expanded by blocks of their own, the first CSSM_SPEC_EXPAND_BLOCKS behind the pack blocks: reading a window costs
// system-scope loads, and an offspring block that expanded its share of the rows behind its own particles ran 5-6 us longer for it
// (in-process shards, 2^20 particles each, tools/archive/expand_cost.py: 11.7 -> 17-17.9 us) -- invisible at world 1, where there are no rows.
// These blocks wait for the headers like every block, then for the neighbours' eager rows, and are done before the offspring blocks are.
#define CSSM_SPEC_EXPAND_BLOCKS 64
__device__ __forceinline__ void expand_spec_body(SpecHeaders& H, uint32_t bid, uint32_t nblk, const double* __restrict__ recv, int world, int rank,
                                                 long long cap, int d, uint32_t n_split, uint64_t slot_lo, uint64_t slot_hi, uint64_t n_global,
                                                 const StepRec* __restrict__ rec, uint32_t* __restrict__ anc, Scalars* __restrict__ sc,
                                                 int rs = CSSM_RESAMPLE_SYSTEMATIC, uint64_t seed = 0, int grp_cur = 0,
                                                 const unsigned int* __restrict__ peer_flags = nullptr, uint32_t peer_seq = 0u, long long eager = 0) {
  // peer_flags (peer-written exchange, behind the neighbours' ROWS flags) and eager < cap: the neighbours wrote the `eager` rows next to
  // the boundary at once and, behind every rank's header, the rows beyond them that this rank's slots need (PackNeed); else every row of
  // the blocks is there
  __shared__ uint32_t s_nheavy;
  __shared__ uint32_t s_hb[CSSM_BLOCK], s_he[CSSM_BLOCK], s_hj[CSSM_BLOCK];
  (void)grp_cur;
  const long long R = d + 1, HD = spec_hdr(d), seg = spec_seg(d, cap);
  const double totd = cssm_u128_to_double(H.tot);
  const double u = rec->u;
  const bool pow2 = (n_global & (n_global - 1)) == 0;
  const double inv_n = 1.0 / (double)n_global;
  const uint32_t rstep = rec->step;
  auto count_of = [&](cssm_u128 G) -> uint64_t {
    if (cssm_u128_is_zero(G)) return 0;
    const double C = cssm_u128_to_double(G) / totd;
    if (rs == CSSM_RESAMPLE_STRATIFIED) return cssm_strat_count(C, seed, rstep, n_global);
    return pow2 ? cssm_sys_count_pow2(C, u, n_global, inv_n) : cssm_sys_count(C, u, n_global);
  };
  // Only the two adjacent ranks' rows can own slots of this rank: the verdict established that its slots below the own
  // particles' first one all belong to the last-cap block of rank - 1 and those above to the first-cap block of rank + 1.
  // The rows are spread evenly over the launch's EXPANSION blocks (bid of nblk: CSSM_SPEC_EXPAND_BLOCKS blocks that do nothing else).
  // EAGER rows first (all of them where nobody asked for less): the last n_lo rows of rank - 1's LAST-cap block, the first n_hi of rank + 1's
  // FIRST-cap block, spread over the blocks
  const long long cnt_lo = (rank > 0) ? H.cnt[rank - 1] : 0, cnt_hi = (rank + 1 < world) ? H.cnt[rank + 1] : 0;
  const long long eg = (peer_flags != nullptr && eager < cap) ? (eager > 0 ? eager : 0) : cap;
  const long long n_lo = (eg < cnt_lo) ? eg : cnt_lo, n_hi = (eg < cnt_hi) ? eg : cnt_hi;
  __shared__ uint32_t s_unc;   // bit 0 / 1: the eager rows of rank - 1 / rank + 1 do not reach this rank's first / last slot
  // rows [idx0, idx1) of the list `which` (0: the eager rows as above; 1 / 2: the rows beyond them of rank - 1 / rank + 1, x_lo of them)
  auto expand_rows = [&](const int which, const long long idx0, const long long idx1, const long long x_n) {
    for (long long base = idx0; base < idx1; base += CSSM_BLOCK) {
      if (threadIdx.x == 0) s_nheavy = 0;
      __syncthreads();
      const long long idx = base + threadIdx.x;
      if (idx < idx1) {
        int s; long long i;
        if (which == 0) { s = (idx < n_lo) ? rank - 1 : rank + 1; i = (idx < n_lo) ? cnt_lo - n_lo + idx : idx - n_lo; }
        else if (which == 1) { s = rank - 1; i = cnt_lo - n_lo - x_n + idx; }
        else { s = rank + 1; i = n_hi + idx; }
        const double* h = recv + (size_t)s * seg;
        const cssm_u128 off = cssm_u128_add(H.off[s], H.base[s]);
        // the row's cumulative weight AND the one in front of it are requested together -- system-scope loads, a round trip each if one
        // waited for the other (the exact counts behind them are chains of their own): the words in front of row i come from row i - 1, or,
        // for the first eager row of rank - 1, from the words that travelled with the rows (PSTART); none for a block's first row and for
        // the first of the rows beyond the eager ones (its run starts at or below this rank's first slot)
        const bool first_eager = which == 0 && s < rank && idx == 0 && i != 0;
        const bool no_prev = i == 0 || (which == 1 && idx == 0);
        const unsigned long long* ps = reinterpret_cast<const unsigned long long*>(peer_flags + (size_t)(first_eager ? s : 0) * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_PSTART);
        const long long ip = (no_prev || first_eager) ? i : i - 1;     // (a row that is there, where none is needed)
        cssm_u128 P, Pp;
        P.lo = cssm_d2u(ld_sys_f64(h + HD + i * R + d)); P.hi = cssm_d2u(ld_sys_f64(h + HD + cap * R + i));
        if (first_eager) {
          Pp.lo = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); Pp.hi = __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          Pp.lo = cssm_d2u(ld_sys_f64(h + HD + ip * R + d)); Pp.hi = cssm_d2u(ld_sys_f64(h + HD + cap * R + ip));
        }
        uint64_t e = count_of(cssm_u128_add(off, P));
        uint64_t b;
        if (i == 0) {
          b = count_of(off);
        } else if (which == 1 && idx == 0) {
          b = slot_lo;   // (the row before the first needed one was not written: its run ends at or below this rank's first slot)
        } else {
          b = count_of(cssm_u128_add(off, Pp));
          // the first eager row's run starts beyond this rank's first slot: rows in front of the eager ones own slots of this rank -- the
          // sender writes them behind the headers (EXTRA)
          if (first_eager && b > slot_lo) atomicOr(&s_unc, 1u);
        }
        // the last eager row of rank + 1 ends below this rank's last slot and the block has more rows: they are on their way (EXTRA)
        if (which == 0 && s > rank && i == n_hi - 1 && n_hi < cnt_hi && e < slot_hi) atomicOr(&s_unc, 2u);
        if (b < slot_lo) b = slot_lo;
        if (e > slot_hi) e = slot_hi;
        if (e > b) {
          const uint32_t row = n_split + (uint32_t)(((long long)s * seg + HD) / R + i);
          if (e - b <= CSSM_RUN_DIRECT) {
            for (uint64_t sl = b; sl < e; ++sl) anc[sl - slot_lo] = row;
          } else {
            const uint32_t hh = atomicAdd(&s_nheavy, 1u);
            s_hb[hh] = (uint32_t)(b - slot_lo); s_he[hh] = (uint32_t)(e - slot_lo); s_hj[hh] = row;
          }
        }
      }
      __syncthreads();
      const uint32_t nh = s_nheavy;
      for (uint32_t hh = 0; hh < nh; ++hh) {
        const uint32_t he = s_he[hh], hj = s_hj[hh];
        for (uint32_t sl = s_hb[hh] + threadIdx.x; sl < he; sl += CSSM_BLOCK) anc[sl] = hj;
      }
      __syncthreads();
    }
  };
  const long long total = n_lo + n_hi;
  const long long per = (total + nblk - 1) / nblk;
  const long long row_lo = (long long)bid * per, row_hi = (row_lo + per < total) ? row_lo + per : total;
  if (row_lo >= row_hi) return;
  if (threadIdx.x == 0) s_unc = 0u;   // (expand_rows begins with a barrier)
  expand_rows(0, row_lo, row_hi, 0);
  if (peer_flags == nullptr || eg >= cap) return;
  // Rows BEYOND the eager ones (rare: the eager rows cover a typical observation several times over).  The block that expanded the first
  // eager row of rank - 1 / the last one of rank + 1 has seen whether they reach; if not it waits, bounded, for the sender's EXTRA flag --
  // set once the sender's row blocks have seen every header and written what this rank needs -- and expands those rows itself.
  const uint32_t unc = s_unc;   // (behind expand_rows' closing barrier)
  if (unc == 0u) return;
  __shared__ unsigned int s_xlate;
  __shared__ long long s_xneed[2];
  __syncthreads();
  if (threadIdx.x < 2) {
    const int side = (int)threadIdx.x, s = side ? rank + 1 : rank - 1;
    s_xneed[side] = 0;
    if (threadIdx.x == 0) s_xlate = 0u;
    if ((unc >> side) & 1u) {
      const unsigned int* f = peer_flags + (size_t)s * CSSM_PEER_FLAG_STRIDE;
      if (!peer_poll_u32(f + CSSM_PEER_FLAG_EXTRA, peer_seq, sc->peer_wait_ticks)) { atomicOr(&s_xlate, 1u); atomicOr(&sc->wait_code, 16u | ((uint32_t)s << 8)); }
      else s_xneed[side] = (long long)__hip_atomic_load(f + CSSM_PEER_FLAG_NEED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (s_xlate) {
    if (threadIdx.x == 0) { atomicOr(&sc->err, 16u); atomicMin(&sc->fail_step, rec->step); }
    return;
  }
  if (unc & 1u) {
    long long need = s_xneed[0]; need = (need < cnt_lo) ? need : cnt_lo;
    if (need > n_lo) expand_rows(1, 0, need - n_lo, need - n_lo);
  }
  if (unc & 2u) {
    long long need = s_xneed[1]; need = (need < cnt_hi) ? need : cnt_hi;
    if (need > n_hi) expand_rows(2, 0, need - n_hi, need - n_hi);
  }
}
// Offspring of the own particles and expansion of the received rows in ONE launch: the two are independent once the
// exchange is done (disjoint slots; both read only the segment headers), and a launch costs ~5 us of latency.  Every block
// is a k_offspring block first and then expands its share of the received rows; `optimistic` = 2 makes the offspring
// side raise err bit 2 itself (no later kernel reads the flag).
// Block 0 takes the verdict of spec_read_headers.  If some rank's slots are not covered, the launch changes NOTHING THAT MATTERS
// on any rank -- no scalars published, no max slots cleared; the ancestor indices the other blocks wrote are overwritten when the
// observation's exchange is redone -- and records err bit 3 and the observation index: every later kernel of the series returns
// at once (they test the bit), and the host redoes this observation's exchange with a larger capacity and carries on
// (cssm_pf_shard_resume).
// RAWC: how the weights are stored, at compile time (2: the weights k_propagate_shard formed relative to the reference level; 0:
// log-weights, rescaled by the level the global max gave -- LGCP, a series repeated after an outlying observation)
// GRP (RAWC == 2; slot_set = the set of Scalars::grp the propagate's blocks added to): the block's prefix INSIDE the rank comes from 32 group
// sums + the 32 unit sums of its own group (as in k_offspring_self<..., GRP>), so nothing is waited for but the peers' headers -- no prefix
// block, no pre_flag -- and everything local (first tile on the grid and scanned, that prefix) is done BEFORE the wait.
template <int RAWC, int RS, bool GRP = false>
__device__ __forceinline__ void offspring_expand_spec_body(
    CSSM_OFFSPRING_PARAMS, uint32_t all5_stride, const double* __restrict__ recv, long long cap, int d, uint32_t n_split,
    const cssm_u128* __restrict__ unit_pre, const unsigned int* __restrict__ peer_flags, uint32_t peer_seq,
    const uint32_t blk0, const unsigned int* __restrict__ pre_flag, SpecHeaders& H, uint32_t* __restrict__ s_ll, const long long eager,
    unsigned int* __restrict__ hdr_done = nullptr) {
  // hdr_done (the merged launch): the count of this launch's header blocks that are done (spec_clear_sets)
  // eager (peer-written exchange): rows next to the boundary that the neighbours wrote at once (PackNeed::eager; >= cap: all of them)
  // H / s_ll: LDS of the launch, declared by the kernel (the merged kernel's pack blocks use the same two objects: static LDS of the
  // two halves of a kernel adds up, it does not overlap)
  // blk0 / pre_flag (the merged kernel k_exchange_offspring): this body runs in blocks blk0 .. of the launch; unit_pre is written by
  // one of the blocks before them and announced through pre_flag
  // blocks blk0 .. blk0 + CSSM_SPEC_EXPAND_BLOCKS - 1 expand the received rows (below), the offspring blocks follow them
  const bool expander = blockIdx.x - blk0 < (uint32_t)CSSM_SPEC_EXPAND_BLOCKS;
  const uint32_t blk1 = blk0 + (uint32_t)CSSM_SPEC_EXPAND_BLOCKS;
  const uint32_t bidx = blockIdx.x - blk1;
  // unit_pre (or nullptr): the exclusive prefixes of the unit sums k_boundary_pack's header block left (its pre_out)
  // peer_flags (peer-written exchange; else nullptr): this rank's flags of the window `recv` is -- one line per source rank; the
  // segment of rank r is complete once its flag holds peer_seq (PeerTable).  Every block waits for every rank's flag (thread r
  // polls rank r's, system-scope acquire, bounded): a rank that never delivers raises err bit 4 (16) here instead of hanging the
  // GPU, and the series ends like one on hold.
  CSSM_SPEC_STAMP(0);
  __shared__ unsigned int s_late;
  const unsigned long long wait_ticks = (peer_flags != nullptr) ? sc->peer_wait_ticks : 0ull;   // (requested with the block's first loads)
  // thread r waits for rank r's flag (word `which` of its pair: 0 header, CSSM_PEER_FLAG_ROWS rows), thread `world` for the local
  // flag `extra` if there is one; false: some flag did not come within the bound (err bit 4, the series ends like one on hold)
  auto wait_flags = [&](uint32_t which, const unsigned int* extra, int r_lo, int r_hi) -> bool {
    if (threadIdx.x == 0) s_late = 0u;
    __syncthreads();
    const int r = (int)threadIdx.x;
    if ((r >= r_lo && r <= r_hi && r < world) || (r == world && extra != nullptr)) {
      const unsigned int* f = (r < world) ? peer_flags + (size_t)r * CSSM_PEER_FLAG_STRIDE + which : extra;
      // (relaxed system-scope loads: each one reads the flag at the point of coherence; the window itself is read with such loads
      //  too -- ld_sys, no fence: see there -- and the next kernel's gathers start behind a kernel boundary)
      if (!peer_poll_u32(f, peer_seq, wait_ticks)) { s_late = 1u; atomicOr(&sc->wait_code, 32u | ((uint32_t)r << 8)); }
    }
    __syncthreads();
    if (s_late) {
      if (threadIdx.x == 0) { atomicOr(&sc->err, 16u); atomicMin(&sc->fail_step, rec->step); }
      return false;
    }
    return true;
  };
  // The neighbours' ROWS flags (their eager rows are complete: they left at the start of the senders' launches, the flags are long set) --
  // thread rank - 1 / rank + 1 polls, one barrier.  s_late is zero here: the header wait that cleared it succeeded.  (Measured and dropped:
  // a look at the flags right behind the headers, kept in a register for the tail -- 95 -> 97 VGPR, a wave of occupancy.)
  const bool rows_thread = peer_flags != nullptr && (int)threadIdx.x < world && ((int)threadIdx.x == rank - 1 || (int)threadIdx.x == rank + 1);
  auto wait_rows = [&]() -> bool {
    if (peer_flags == nullptr) return true;
    if (rows_thread &&
        !peer_poll_u32(peer_flags + (size_t)threadIdx.x * CSSM_PEER_FLAG_STRIDE + CSSM_PEER_FLAG_ROWS, peer_seq, wait_ticks)) {
      s_late = 1u; atomicOr(&sc->wait_code, 8u | (threadIdx.x << 8));
    }
    __syncthreads();
    if (s_late) {
      if (threadIdx.x == 0) { atomicOr(&sc->err, 16u); atomicMin(&sc->fail_step, rec->step); }
      return false;
    }
    return true;
  };
  if (expander) __builtin_amdgcn_s_setprio(1);   // (few blocks with a chain of their own: ahead of the offspring blocks they share a CU with)
  else if (bidx == 0) __builtin_amdgcn_s_setprio(2);   // (the block with the verdict and the publication on top of its unit)
  if (expander) {
    // an expansion block: every rank's header, the level test (block 0 of the offspring blocks records a ruled-out level; nobody expands
    // then), the neighbours' eager rows, its share of the rows.  (The coverage verdict stays the first offspring block's business: rows
    // expanded into slots of an exchange that is redone are written again.)
    if (sc->err & (4u | 8u | 16u)) return;   // (on hold / void / a peer missing: nobody delivers, nobody waits)
    const double rec_ref = rec->ref;
    SpecHdrRegs hregs;
    const bool xstamp = peer_flags != nullptr && blockIdx.x == blk0;   // (the first expansion block keeps the time of its waits: Scalars::xhdr_wait_ticks)
    const unsigned long long tx0 = xstamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (peer_flags != nullptr) {
      if constexpr (GRP) {
        if (!peer_headers_ll(hregs, s_ll, s_late, peer_flags, peer_seq, wait_ticks, world)) {
          if (threadIdx.x == 0) { atomicOr(&sc->err, 16u); atomicMin(&sc->fail_step, rec->step); atomicOr(&sc->wait_code, 2u | (s_late & 0xff00u)); }
          return;
        }
      } else {
        if (!wait_flags(0u, nullptr, 0, world - 1)) return;
        hregs = spec_load_headers(recv, world, cap, d);
      }
    } else {
      hregs = spec_load_headers(recv, world, cap, d);
    }
    spec_store_headers(H, hregs, world, cap);
    if (optimistic && !(cssm_ref_choose(rec_ref, cssm_order_unkey(H.gkey)) == rec_ref)) return;
    const unsigned long long tx1 = xstamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (!wait_rows()) return;
    if (xstamp && threadIdx.x == 0) { sc->xhdr_wait_ticks += tx1 - tx0; sc->rows_wait_ticks += __builtin_amdgcn_s_memrealtime() - tx1; }
    expand_spec_body(H, blockIdx.x - blk0, (uint32_t)CSSM_SPEC_EXPAND_BLOCKS, recv, world, rank, cap, d, n_split, (uint64_t)slot_lo, (uint64_t)slot_hi, n_global, rec, anc, sc,
                     RS, seed, slot_set, peer_flags, peer_seq, eager);
    CSSM_SPEC_STAMP(3);
    return;
  }
  double pre_w[CSSM_ITEMS];
  if constexpr (GRP) {
    static_assert(RAWC == 2, "group sums: behind a propagate that formed the sums");
    const bool pre_ok = bidx < nunits;                       // (uniform; a launch with the group sums has exactly nunits offspring blocks)
    if (pre_ok) load_tile_raw(logw, (uint64_t)bidx * sup * CSSM_TILE, n, RAWC, pre_w);
    const uint32_t held0 = sc->err;
    const double rec_ref = rec->ref, rec_u = rec->u;
    if (held0 & (4u | 8u | 16u)) return;                     // (on hold / void / a peer missing: nobody delivers, nobody waits)
    bool mid_ok = false;                                     // (the body called mid and it said yes: this block resampled)
    auto mid = [&](SpecTotals& tt) -> bool {
      SpecHdrRegs hregs;
      if (peer_flags != nullptr) {
        const unsigned long long tw0 = (bidx == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (!peer_headers_ll(hregs, s_ll, s_late, peer_flags, peer_seq, wait_ticks, world)) {
          if (threadIdx.x == 0) { atomicOr(&sc->err, 16u); atomicMin(&sc->fail_step, rec->step); atomicOr(&sc->wait_code, 1u | (s_late & 0xff00u)); }
          return false;
        }
        if (bidx == 0 && threadIdx.x == 0) { sc->hdr_wait_ticks += __builtin_amdgcn_s_memrealtime() - tw0; sc->xwaits += 1ull; }   // (Scalars::hdr_wait_ticks)
        CSSM_SPEC_STAMP(7);
      } else {
        hregs = spec_load_headers(recv, world, cap, d);
      }
      spec_store_headers(H, hregs, world, cap);
      if (optimistic && !(cssm_ref_choose(rec_ref, cssm_order_unkey(H.gkey)) == rec_ref)) {   // (see below: the level first)
        if (bidx == 0 && threadIdx.x == 0) { atomicMin(&sc->fail_step, rec->step); atomicOr(&sc->err, 4u); }
        return false;
      }
      if (bidx == 0 && !spec_read_headers(H, world, rank, cap, d, n, n_global, rec_u, RS, seed, rec->step)) {   // (block 0's business alone: see below)
        if (threadIdx.x == 0) { atomicOr(&sc->err, 8u); atomicMin(&sc->fail_step, rec->step); }
        return false;
      }
      CSSM_SPEC_STAMP(1);
      tt.S_off.lo = H.off[rank].lo; tt.S_off.hi = H.off[rank].hi; tt.tot.lo = H.tot.lo; tt.tot.hi = H.tot.hi;
      tt.tot2.lo = H.tot2.lo; tt.tot2.hi = H.tot2.hi; tt.gmax = cssm_order_unkey(H.gkey);
      mid_ok = true;
      return true;
    };
    (void)raw; (void)unit_pre; (void)pre_flag;
    offspring_body<true, false, RS, RAWC, true, decltype(mid)>(logw, n, sc, unitP, unitS2, rec, n_global, /*endslot=*/nullptr, anc, ntiles, sup, nunits, RAWC, slot_set,
                                                               /*ll_t=*/nullptr, /*ess_t=*/nullptr, 0u, force_exact, all5, rank, world, split, seed,
                                                               /*cum_out=*/nullptr, /*logtab=*/nullptr, optimistic, flag_out, slot_lo, slot_hi, all5_stride,
                                                               /*s2buf=*/nullptr, 0u, -1, 0u, /*unit_pre=*/nullptr, blk1, pre_ok ? pre_w : nullptr, nullptr, &mid);
    if (!mid_ok) return;   // (the body returned without resampling -- the level ruled out, block 0's verdict, a peer missing: no tail either)
    CSSM_SPEC_STAMP(2);
    if (bidx == 0) spec_clear_sets(sc, slot_set, hdr_done, (unsigned int)world);
    return;
  }
  const bool prefetched = peer_flags != nullptr && bidx < nunits;   // (uniform)
  if (peer_flags != nullptr) {
    // the block's first tile of weights is requested BEFORE the wait (it depends on nothing the peers send)
    if (prefetched) load_tile_raw(logw, (uint64_t)bidx * sup * CSSM_TILE, n, RAWC, pre_w);
    if (sc->err & (4u | 8u | 16u)) return;   // (on hold / void / a peer missing: nobody delivers, nobody waits)
    if (!wait_flags(0u, pre_flag, 0, world - 1)) return;          // every rank's header (and this launch's unit-sum prefixes)
    CSSM_SPEC_STAMP(7);
  }
  // everything the verdict starts from is requested first, tested afterwards
  const uint32_t held = sc->err;
  const double rec_ref = rec->ref, rec_u = rec->u;
  const SpecHdrRegs hregs = spec_load_headers(recv, world, cap, d);   // (thread r: the 12 header words of rank r -- sums, max key, block totals)
  // (measured and dropped: the weights of the block's first tile requested here as well, ahead of the verdict -- 11.8 -> 12.3 us at
  //  2^20 per rank, eight more registers live across the verdict)
  if (held & (4u | 8u | 16u)) return;
  spec_store_headers(H, hregs, world, cap);
  if (optimistic) {
    // the level first: sums formed relative to a reference level that the global max rules out (an outlying observation) say
    // nothing about coverage either.  Sticky bit 2 (4): every later kernel of the series returns at once, the host runs
    // the series again with the levels taken from the global max.
    if (!(cssm_ref_choose(rec_ref, cssm_order_unkey(H.gkey)) == rec_ref)) {
      if (bidx == 0 && threadIdx.x == 0) { atomicMin(&sc->fail_step, rec->step); atomicOr(&sc->err, 4u); }   // (the series holds HERE: cssm_pf_shard_resume_level)
      return;
    }
  }
  // The coverage verdict is BLOCK 0's business alone.  What must not happen on a capacity miss is what block 0 does: publishing the
  // observation's scalars (ll would count twice when the observation is redone) and clearing the max slots (the redone pack reads
  // them again).  The other blocks write nothing but ancestor indices -- of slots the redone exchange writes again, all of them, before
  // anybody gathers through them -- so they do not wait for a verdict they cannot act on: three block barriers and 4 x world exact
  // slot counts less on the path of every block but one (round 3: 2.9 us from entry to verdict in all 1024 blocks).
  if (bidx == 0 && !spec_read_headers(H, world, rank, cap, d, n, n_global, rec_u, RS, seed, rec->step)) {
    if (threadIdx.x == 0) { atomicOr(&sc->err, 8u); atomicMin(&sc->fail_step, rec->step); }
    return;
  }
  CSSM_SPEC_STAMP(1);
  // the ranks' totals as the body wants them: read once per block above, not world x 5 times per thread
  SpecTotals tt;
  tt.S_off.lo = H.off[rank].lo; tt.S_off.hi = H.off[rank].hi; tt.tot.lo = H.tot.lo; tt.tot.hi = H.tot.hi;
  tt.tot2.lo = H.tot2.lo; tt.tot2.hi = H.tot2.hi; tt.gmax = cssm_order_unkey(H.gkey);
  // (the arguments the single-collective launch has no use for are constants here: the compiler drops what hangs on them)
  (void)raw;
  offspring_body<true, false, RS, RAWC>(logw, n, sc, unitP, unitS2, rec, n_global, /*endslot=*/nullptr, anc, ntiles, sup, nunits, RAWC, slot_set,
                                                        /*ll_t=*/nullptr, /*ess_t=*/nullptr, 0u, force_exact, all5, rank, world, split, seed,
                                                        /*cum_out=*/nullptr, /*logtab=*/nullptr, optimistic, flag_out, slot_lo, slot_hi, all5_stride,
                                                        /*s2buf=*/nullptr, 0u, -1, 0u, unit_pre, blk1, prefetched ? pre_w : nullptr, &tt);
  CSSM_SPEC_STAMP(2);
  if (bidx == 0) spec_clear_sets(sc, slot_set, hdr_done, (unsigned int)world);
}

template <int RAWC, int RS = CSSM_RESAMPLE_SYSTEMATIC, bool GRP = false>
__global__ __launch_bounds__(CSSM_BLOCK, CSSM_OFF_WAVES) void k_offspring_expand_spec(
    CSSM_OFFSPRING_PARAMS, uint32_t all5_stride, const double* __restrict__ recv, long long cap, int d, uint32_t n_split,
    const cssm_u128* __restrict__ unit_pre, const unsigned int* __restrict__ peer_flags = nullptr, uint32_t peer_seq = 0u, long long eager = 0) {
  __shared__ SpecHeaders H;
  __shared__ uint32_t s_ll[GRP ? 64 * CSSM_PEER_LL_WORDS : 1];   // the ranks' headers as they arrive: 24 halves per rank (GRP launches read them)
  offspring_expand_spec_body<RAWC, RS, GRP>(CSSM_OFFSPRING_FWD, all5_stride, recv, cap, d, n_split, unit_pre, peer_flags, peer_seq, 0u, nullptr, H, s_ll, eager);
}

// The peer-written exchange in ONE launch per weighted observation behind the propagate: the first pack_gx * world blocks of the grid
// are k_boundary_pack's -- they write this rank's segments into the peers' windows, set the flags and leave (they wait for nothing,
// and being the first blocks of the grid they are dispatched first: nobody who polls can keep them from running) -- and every
// later block is a k_offspring_expand_spec block: it waits for all ranks' flags (and for the unit-sum prefixes one of the pack
// blocks leaves), then resamples.  A dependent launch less per observation (~3 us of launch + the pack kernel's own first round trips).
struct PackArgs {
  const double* src; size_t stride; uint32_t nsub; uint64_t chunk; cssm_u128* pre_out;
  const PeerTable* peer; int parity; unsigned int* tickets; unsigned int* pre_flag; uint32_t pack_gx;
  long long eager;   // rows next to the boundary that travel at once (PackNeed::eager; >= cap: every row, CSSM_PEER_ALL_ROWS)
};
// GRP: the group sums are at hand (slot_set = their set): the header blocks total them, the offspring blocks take their prefixes from them
// (pk.pre_out == nullptr: the prefix block of every destination leaves at once)
template <int RAWC, int RS = CSSM_RESAMPLE_SYSTEMATIC, bool GRP = false>
__global__ __launch_bounds__(CSSM_BLOCK, CSSM_OFF_WAVES) void k_exchange_offspring(
    CSSM_OFFSPRING_PARAMS, uint32_t all5_stride, const double* __restrict__ recv, long long cap, int d, uint32_t n_split,
    const cssm_u128* __restrict__ unit_pre, const unsigned int* __restrict__ peer_flags, uint32_t peer_seq, PackArgs pk) {
  const uint32_t blk0 = pk.pack_gx * (uint32_t)world;
  __shared__ SpecHeaders H;
  __shared__ uint32_t s_ll[64 * CSSM_PEER_LL_WORDS];   // the ranks' headers as they arrive: 24 halves per rank
  if (blockIdx.x < blk0) {
    PackNeed xn;
    xn.H = &H; xn.ll = s_ll; xn.my_flags = peer_flags; xn.need = pk.tickets + CSSM_PEER_TICKET_NEED;
    xn.stat = reinterpret_cast<unsigned long long*>(pk.tickets + CSSM_PEER_TICKET_STAT);
    xn.n_global = n_global; xn.seed = seed; xn.slot_lo = slot_lo; xn.slot_hi = slot_hi; xn.rs = RS; xn.eager = pk.eager;
    boundary_pack_block<RS>(blockIdx.x % pk.pack_gx, pk.pack_gx, (int)(blockIdx.x / pk.pack_gx), pk.src, pk.stride, logw, n, d, world, rank, cap, rec,
                        unitP, unitS2, pk.nsub, sc, nullptr, pk.chunk, /*level_from_max=*/0, pk.pre_out, pk.peer, pk.parity, peer_seq, pk.tickets,
                        pk.pre_flag, GRP ? slot_set : -1, xn, pk.eager < cap, 7, /*merged=*/true);
    return;
  }
  offspring_expand_spec_body<RAWC, RS, GRP>(CSSM_OFFSPRING_FWD, all5_stride, recv, cap, d, n_split, unit_pre, peer_flags, peer_seq, blk0, pk.pre_flag, H, s_ll, pk.eager,
                                            pk.tickets + CSSM_PEER_TICKET_HDR_DONE);
}

__global__ __launch_bounds__(CSSM_BLOCK) void k_expand(const uint32_t* __restrict__ cand_end, const uint32_t* __restrict__ cand_idx,
                                                       uint64_t m, uint64_t n_low, uint64_t slot_lo, uint64_t slot_hi,
                                                       uint32_t* __restrict__ anc, const uint32_t* __restrict__ own_last_end) {
  expand_body(cand_end, cand_idx, m, n_low, slot_lo, slot_hi, anc, own_last_end);
}

// cssm_internal.h -- what the HIP translation units of libcssm_pf share: the handle, error plumbing and the launch helpers
// that both the single-GPU drivers (cssm_pf.hip) and the sharded stages (cssm_shard.hip) use.  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "cssm_host.h"

#define fail cssm_fail
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) return fail(CSSM_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define CSSM_NKERNELS CSSM_PROFILE_NKERNELS


struct cssm_pf : HostModel {
  int device = 0;
  int n_cus = 256;             // compute units of the device (launch geometry decisions)
  hipStream_t stream = nullptr;
  bool own_stream = true;
  // sizes
  uint64_t first = 0, n = 0;   // (n_global, seed: HostModel)
  size_t stride = 0;
  uint32_t ntiles = 0;
  uint32_t sup = 1, nunits = 0;   // tiles per scan unit, number of units (<= ~1K)
  uint32_t split = 1;             // k_propagate blocks per unit (each owns a contiguous sub-unit and its sums)
  bool safe_sums = false;         // form the sums in their own pass after the max is known (retry of a step whose
                                  // reference level was ruled out by the max; always for LGCP)
  bool last_optimistic = false;   // the last launch_propagate formed the sums itself (and stored weights in place of log-weights)
  bool wmode = false;             // the log-weight buffer holds the WEIGHTS exp(min(w - c, 2^-20)) of the last weighted observation
  bool have_level = false;        // LGCP (contract v8): a weighted observation has run natively since the cloud was drawn / adopted -- its
                                  //   publisher left the next observation's predicted level (Scalars::next_ref), so the propagate may form the sums
  bool sums_ready = false;        // sharded: the unit sums of the last propagated observation exist (k_propagate<SUMS> or cssm_pf_shard_sums)
  // ESS pending (Scalars::pend): k_offspring's blocks leave partial sums of squared weights in s2buf[par * s2_stride + block]
  cssm_u128* s2buf = nullptr;     // 2 x s2_stride entries
  uint32_t s2_stride = 0;
  int s2_par = 0;                 // buffer the next weighted observation's k_offspring writes
  uint32_t gen = 0;               // generation of the batch call / streaming step that owns d_ess_t
  int32_t ess_host = 0;           // ESS of the last completed observation as the host knows it
  void* last_comm = nullptr;      // RCCL communicator the library last enqueued collectives on (bounded_sync)
  std::vector<double> h_fsub;     // host copy of the sub-step coefficient table of the records last built
  double* d_fsub = nullptr; size_t fsub_cap = 0;
  bool sharded = false;
  // device memory
  double* state[2] = {nullptr, nullptr};
  int cur = 0;                 // state[cur] = propagated cloud of the last step (x1)
  const double* src = nullptr; // where the next propagate reads (state[cur])
  size_t src_stride = 0;
  const double* src2 = nullptr; // sharded: candidates received from other ranks (indices >= n_split)
  size_t src2_stride = 0;
  uint32_t n_split = 0;
  double* logw = nullptr;
  uint32_t* endslot = nullptr;
  uint32_t* anc = nullptr;
  bool anc_valid = false;
  int wparity = 0;             // max-slot set (0 .. CSSM_MAXSETS - 1) of the next weighted step (single-GPU path); a shard: the set of GROUP
                               //   sums (Scalars::grp / grp2) of the next weighted observation -- its max slots stay in set 0
  int opt_exact = 0;           // CSSM_OPT_EXACT_OFFSPRING
  int opt_fused = 0;           // CSSM_OPT_FUSED_SUMS (set to 1 for sharded handles at creation)
  cssm_u128 *fineS = nullptr, *fineS2 = nullptr;   // large clouds: the sums of k_propagate's single-tile blocks (k_reduce_units folds them into tileS / tileS2)
  size_t fine_cap = 0;
  int opt_whole = 0;           // CSSM_OPT_WHOLE_TILES
  int opt_grp = 1;             // CSSM_OPT_GROUP_SUMS
  uint32_t grp_min_units = 2u * CSSM_GRP_UNITS;   // smallest cloud (in units) whose propagate accumulates group sums (CSSM_GRP_MIN_UNITS: tests run small clouds through them)
  int opt_spec = 1;            // CSSM_OPT_SPECIALISE
  bool last_grp = false;       // the last launch_propagate's blocks accumulated the sums of groups of units (Scalars::grp)
  int grp_layout = 0;          // ... in layout 1 (<= 32 groups of 32 units) or 2 (<= 64 groups of 64 units); 0: not at all
  bool last_ws = false;        // ... and its waves stored the exact sums of their quarter units (tileW): k_offspring_wave may run
  int opt_wave = 1;            // CSSM_OPT_WAVE_SUMS
  cssm_u128* tileW = nullptr;  // four per unit: the quarter units' sums (k_propagate_self<..., SUMS = 1, ONE = 2>, bit 13 of its set argument)
  int resampler = CSSM_RESAMPLE_SYSTEMATIC;
  double* cum = nullptr;       // multinomial: cumulative normalised weights
  const long long *send_first_dev = nullptr, *send_count_dev = nullptr;   // last shard_offspring outputs (device)
  cssm_u128 *tileS = nullptr, *tileS2 = nullptr, *tileP = nullptr;
  cssm_u128* unitPre = nullptr;   // sharded, single-collective exchange: exclusive prefixes of the unit sums (k_boundary_pack -> k_offspring_expand_spec)
  bool spec_pre = false;          // ... written by the last cssm_pf_shard_boundary_pack
  Scalars* sc = nullptr;
  double *d_m0 = nullptr, *d_sd0 = nullptr, *d_logtab = nullptr;
  StepRec* d_recs = nullptr;
  size_t recs_cap = 0;
  double* d_ll_t = nullptr;
  int32_t* d_ess_t = nullptr;
  // host-mapped mirrors k_finish writes at the end of a call (the host synchronises the stream and reads them: no D2H copy)
  double* h_ll_t = nullptr; double* hd_ll_t = nullptr;       // host pointer / the same memory as the device sees it
  int32_t* h_ess_t = nullptr; int32_t* hd_ess_t = nullptr;
  Scalars* hd_sc = nullptr;                                   // device view of h_sc
  double* d_path = nullptr;
  size_t path_cap = 0;
  // sharded extras
  double* cand = nullptr;      // candidate states received for this rank, SoA [d][cand_cap]
  size_t cand_cap = 0;
  uint32_t *cand_end = nullptr, *cand_idx = nullptr;   // end slot / state index of every candidate, global order
  size_t cidx_cap = 0;
  int64_t* d_bounds = nullptr;
  int64_t* d_xch = nullptr;    // exact exchange: [0..63] send first, [64..127] send count; [128] the redo flag k_offspring_expand_spec writes
  uint32_t* d_need = nullptr;  // per observation: rows the exchange needed (diagnostics of cssm_pf_shard_status; zero since the
  size_t need_cap = 0;         //   two-collective exchange that recorded them was removed)
  bool series = false;         // records of a whole series are resident (cssm_pf_shard_begin / _continue)
  // sharded cloud summaries (cssm_pf_shard_summary_*): keys of the local cloud, block partial sums, the two radix-select states per row
  unsigned long long* sm_keys = nullptr; double* sm_partial = nullptr; void* sm_st = nullptr; StepRec* sm_rec = nullptr;
  size_t sm_cap = 0; int sm_blocks = 0; double sm_time = 0.0;
  // peer-written exchange (cssm_shard.hip: PeerState; the device table and the ticket counters k_boundary_pack uses)
  void* peer = nullptr; void* peer_tab = nullptr; unsigned int* peer_tickets = nullptr;
  unsigned long long peer_wait_ticks = 3000000000ull;   // 30 s of the 100 MHz clock (CSSM_PEER_TIMEOUT_MS): Scalars::peer_wait_ticks
  uint32_t peer_seq = 0;       // number of the last exchange enqueued on the peer windows (window = seq & 1; the flags carry it)
  uint32_t peer_probe_plain_bad = 0;   // cssm_pf_shard_peer_probe_stale
  bool peer_packed = false;    // cssm_pf_shard_pack_peer ran, cssm_pf_shard_adopt_peer has not yet
  bool peer_rows_packed = false;   // ... and cssm_pf_shard_pack_rows_peer (the second stage of a pack that sends the needed rows only)
  bool peer_all_rows = false;  // CSSM_PEER_ALL_ROWS=1: every row of the boundary blocks travels (else the eager rows + what the neighbour's slots need beyond them)
  long long peer_eager = 4096; // CSSM_PEER_EAGER_ROWS: rows next to the boundary that travel at once, ahead of the headers (four tiles)
  bool want_path = false;      // sharded `filter`: record sampleOne's pick after every observation whose slot this rank owns
  uint32_t rec_base = 0;       // observation index (pf->step) of the resident series' first record: 0 after _begin, the filter's
                               //   observation count so far after _continue
  struct Snap { int cur; const double* src; size_t src_stride; const double* src2; size_t src2_stride; uint32_t n_split; bool anc_valid, last_optimistic; uint32_t step; double t; bool have_level;
                int wparity; bool last_grp; };   // (the set of group sums the observation's propagate added to, and whether it did)
  std::vector<Snap> snaps;     // host-side state right after the propagate of every observation of the series (cssm_pf_shard_resume)
  std::vector<Snap> pre_snaps; // ... and right BEFORE it (cssm_pf_shard_resume_level: the observation is propagated again)
  // host staging (pinned)
  StepRec* h_recs = nullptr;
  StepRec* h_recs_dev = nullptr;   // the device's address of h_recs (k_fetch_recs)
  size_t h_recs_cap = 0;
  Scalars* h_sc = nullptr;     // pinned: the streaming step's scalars land here without a staging copy
  // filter state
  double t = 0.0;
  uint32_t step = 0;
  uint32_t h_step_for_resample = 0;   // observation index of the step being resampled (Philox counter word)
  bool initialised = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = -1.f;
  int opt_events = 0;              // CSSM_OPT_LOOP_EVENTS: the batch drivers bracket their device loop with an event pair
  bool have_events = false;        // ... and the last call did
  // optional per-kernel timing (HIP events on the launch stream around every kernel)
  bool profile = false;
  uint32_t* h_done = nullptr; uint32_t* hd_done = nullptr; uint32_t done_seq = 0;   // k_finish's completion word (host-mapped), read_scalars polls it
  std::vector<hipEvent_t> prof_ev;         // pairs
  std::vector<int> prof_kind;              // kernel kind of pair i
  size_t prof_used = 0;
  double prof_ms[CSSM_NKERNELS] = {0};
  uint64_t prof_cnt[CSSM_NKERNELS] = {0};
};

// begin/end of one profiled launch
static inline void prof_begin(cssm_pf* pf, int kind) {
  if (!pf->profile) return;
  if (pf->prof_used * 2 + 2 > pf->prof_ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { pf->profile = false; return; }
    pf->prof_ev.push_back(a); pf->prof_ev.push_back(b);
  }
  if (pf->prof_kind.size() <= pf->prof_used) pf->prof_kind.resize(pf->prof_used + 1);
  pf->prof_kind[pf->prof_used] = kind;
  (void)hipEventRecord(pf->prof_ev[pf->prof_used * 2], pf->stream);
}
static inline void prof_end(cssm_pf* pf) {
  if (!pf->profile) return;
  (void)hipEventRecord(pf->prof_ev[pf->prof_used * 2 + 1], pf->stream);
  pf->prof_used++;
}
// after a stream synchronise: fold the recorded pairs into per-kind totals
static inline void prof_collect(cssm_pf* pf) {
  for (size_t i = 0; i < pf->prof_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pf->prof_ev[2 * i], pf->prof_ev[2 * i + 1]) == hipSuccess) {
      pf->prof_ms[pf->prof_kind[i]] += ms;
      pf->prof_cnt[pf->prof_kind[i]]++;
    }
  }
  pf->prof_used = 0;
}

static inline int grid_for(uint64_t n, int block, int cap) {
  uint64_t g = (n + block - 1) / block;
  if (g > (uint64_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// The host reads the scalars BEHIND the max slots (err, ess, fail_step, gmax, ref, ll, the sums: ~120 bytes, not the 24 KiB of
// slot lines in front of them -- that copy to pageable memory cost ~15 us per streaming step and per batch call).
#define CSSM_SC_TAIL_OFF offsetof(Scalars, err)
#define CSSM_SC_TAIL_ARGS(hp, scp) reinterpret_cast<char*>(hp) + CSSM_SC_TAIL_OFF, reinterpret_cast<const char*>(scp) + CSSM_SC_TAIL_OFF, sizeof(Scalars) - CSSM_SC_TAIL_OFF

static const int kGridCap = 4096;

#define DISPATCH_D(d, ...)                                                         \
  switch (d) {                                                                      \
    case 1: { constexpr int D = 1; __VA_ARGS__; } break;   case 2: { constexpr int D = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int D = 3; __VA_ARGS__; } break;   case 4: { constexpr int D = 4; __VA_ARGS__; } break;   \
    case 5: { constexpr int D = 5; __VA_ARGS__; } break;   case 6: { constexpr int D = 6; __VA_ARGS__; } break;   \
    case 7: { constexpr int D = 7; __VA_ARGS__; } break;   case 8: { constexpr int D = 8; __VA_ARGS__; } break;   \
    case 9: { constexpr int D = 9; __VA_ARGS__; } break;   case 10: { constexpr int D = 10; __VA_ARGS__; } break; \
    case 11: { constexpr int D = 11; __VA_ARGS__; } break; case 12: { constexpr int D = 12; __VA_ARGS__; } break; \
    case 13: { constexpr int D = 13; __VA_ARGS__; } break; case 14: { constexpr int D = 14; __VA_ARGS__; } break; \
    case 15: { constexpr int D = 15; __VA_ARGS__; } break; default: { constexpr int D = 16; __VA_ARGS__; } break; \
  }

// ---- defined in cssm_pf.hip, used by cssm_shard.hip as well
int cssm_build_fsub(cssm_pf* pf, size_t first, size_t count, bool reset);   // sub-step table of records [first, +count) -> device
int cssm_ensure_recs(cssm_pf* pf, size_t T);
int cssm_upload_recs(cssm_pf* pf, size_t first, size_t count, bool chain);   // records -> device (+ the predicted level of the first one: LGCP)
int cssm_launch_init(cssm_pf* pf, double t0);
// batched launch (cssm_batch.hip): the chains of the launch; `pf` is chain 0, whose bookkeeping stands for all of them
struct CssmBatchLaunch { const void* chains; int nchains; uint32_t rec_idx; int want_pick; };
int cssm_launch_propagate(cssm_pf* pf, const StepRec* d_rec, double* pick_out = nullptr, uint32_t pick_slot = 0, const CssmBatchLaunch* batch = nullptr);
int cssm_pf_create_on_stream(const cssm_model_desc* desc, uint64_t n_particles, uint64_t seed, int device, hipStream_t stream, cssm_pf** out);
void cssm_batch_fresh(cssm_pf* pf, double t0);   // host-side state of a freshly drawn cloud; the call's generation and completion number
int cssm_batch_ok(const cssm_pf* pf);            // the batched launches serve this handle's configuration
int cssm_check_device_err(cssm_pf* pf, const Scalars& h);
int cssm_prop_items(int d);   // PropItems<D>
double cssm_eta_of_mean(const cssm_pf* pf, const StepRec& rec, const double* mean);
void cssm_peer_free(cssm_pf* pf);   // cssm_shard.hip: the peer-written exchange's windows, mappings and device table   // link(f(stateMean, t)), ParticleFilter.scala:420

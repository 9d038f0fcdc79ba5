// cssm_pf.hip -- the single-GPU drivers of libcssm_pf: handle lifetime, kernel launches, the streaming and batch entry points of
// include/cssm_pf.h, inspection, cloud summaries, FilterInterpolate, the stateless resampler, diagnostics.  No torch, no CPU
// compute path: every entry point drives HIP.  Host-only model code lives in cssm_model.cpp, the sharded stages and the RCCL
// series loop in cssm_shard.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "cssm_internal.h"
#include "cssm_kernels.hip.h"
#include "cssm_offspring_wave.hip.h"

static int upload_init_params(cssm_pf* pf) {
  double m0[CSSM_MAX_DIM], sd0[CSSM_MAX_DIM];
  for (int k = 0; k < pf->d; ++k) { m0[k] = pf->comp[k].m0; sd0[k] = std::sqrt(pf->comp[k].c0); }
  HIP_TRY(hipMemcpyAsync(pf->d_m0, m0, pf->d * 8, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(pf->d_sd0, sd0, pf->d * 8, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

// The sub-step coefficient table of records [first, first + count) (LGCP with a time-dependent f; cssm_model.cpp builds it),
// uploaded behind the records.  `reset`: the records start a new table.
int cssm_build_fsub(cssm_pf* pf, size_t first, size_t count, bool reset) {
  if (!pf->lgcp_tdep) return CSSM_OK;
  if (reset) pf->h_fsub.clear();
  int rc = cssm_build_fsub_table(pf, pf->h_recs, first, count, pf->h_fsub);
  if (rc) return rc;
  if (pf->h_fsub.empty()) return CSSM_OK;
  if (pf->fsub_cap < pf->h_fsub.size()) {
    HIP_TRY(hipStreamSynchronize(pf->stream));
    if (pf->d_fsub) (void)hipFree(pf->d_fsub);
    pf->d_fsub = nullptr; pf->fsub_cap = 0;
    const size_t cap = pf->h_fsub.size() + pf->h_fsub.size() / 2 + 1024;
    HIP_TRY(hipMalloc(&pf->d_fsub, cap * 8));
    pf->fsub_cap = cap;
  }
  // (synchronous: the source is an ordinary vector that the next call may rewrite; this happens once per series, or once
  //  per observation in the streaming calls, which wait for the device per observation anyway)
  HIP_TRY(hipStreamSynchronize(pf->stream));
  HIP_TRY(hipMemcpy(pf->d_fsub, pf->h_fsub.data(), pf->h_fsub.size() * 8, hipMemcpyHostToDevice));
  return CSSM_OK;
}

// ------------------------------------------------------------------------------------ create / destroy

#ifndef CSSM_PROP_IT_LO
#define CSSM_PROP_IT_LO 2
#endif
#ifndef CSSM_SPLIT_MAX_N
#define CSSM_SPLIT_MAX_N (1u << 20)
#endif
int cssm_prop_items(int d) { return d <= 2 ? CSSM_PROP_IT_LO : (d <= 8 ? CSSM_PROP_IT_MID : 1); }   // PropItems<D>

// k_propagate blocks per 1024-particle unit on a single-GPU handle: one tile of the kernel per block below 2^20 particles
static uint32_t auto_split(const cssm_pf* pf) {
  // (measured and dropped: a shard of an LGCP filter on blocks of 1024 particles where a unit has 2048 -- k_propagate 71.8 -> 68.4 us at
  //  2^21 particles per rank, one round of whole-unit blocks running in lockstep being 12 % slower per particle than the same kernel on
  //  several rounds; but the exchange kernel's header and prefix blocks, which every block of every rank waits for, then total twice
  //  the sums: 18.5 -> 20.4 us, the step 89.5 -> 90.3)
  if (pf->sharded) {
    // an LGCP shard whose units have several tiles: blocks of HALF a unit -- whole-unit blocks all start together (1024 of them, one
    // round) and run in lockstep, so every memory wait of the prologue and of the tile boundaries is exposed; half-unit blocks run in
    // two staggered rounds (k_propagate 71.8 -> 68.4 us at 2^21 particles per rank).  The exchange kernels do not see the difference
    // any more: they read the sums of groups of units (Scalars::grp), to which every block adds its own.  CSSM_SHARD_LGCP_SPLIT: 1, 2, 4.
    if (pf->obs_kind != CSSM_OBS_LGCP || pf->sup < 2) return 1u;
    uint32_t want = 2u;
    if (const char* e = getenv("CSSM_SHARD_LGCP_SPLIT")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) want = (uint32_t)v; }
    // (whole tiles of 1024: the exchange's pack blocks take the prefix of a boundary block's tiles from the blocks' sums only then --
    //  quarter-unit blocks of 512 particles at 2^21 per rank sent them down their recompute path: 39 us per exchange)
    while (want > 1u && ((uint64_t)pf->sup * CSSM_TILE / want) % (uint64_t)CSSM_TILE != 0) want >>= 1;
    const uint64_t unit_particles = (uint64_t)pf->sup * CSSM_TILE;
    if (!(pf->opt_grp && pf->nunits >= pf->grp_min_units && pf->nunits <= (uint32_t)(CSSM_GRP_SMALL * CSSM_GRP_UNITS) && unit_particles <= CSSM_GRP_MAX_UNIT)) want = 1u;   // (no group sums: whole units)
    return want;
  }
  if (pf->sup != 1 || pf->n >= CSSM_SPLIT_MAX_N) return 1u;
  return cssm_prop_items(pf->d) == 1 ? 4u : 2u;   // one tile of the kernel: half of 1024 (two particles per thread), a quarter (one)
}

static int alloc_handle(cssm_pf* pf) {
  HIP_TRY(hipSetDevice(pf->device));
  if (pf->own_stream) HIP_TRY(hipStreamCreateWithFlags(&pf->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&pf->ev0));
  HIP_TRY(hipEventCreate(&pf->ev1));
  { const char* e = getenv("CSSM_LOOP_EVENTS"); pf->opt_events = (e && e[0] == '1') ? 1 : 0; }   // (measurement scripts: the default of CSSM_OPT_LOOP_EVENTS)
  { const char* e = getenv("CSSM_PEER_TIMEOUT_MS"); if (e && atof(e) > 0.0) pf->peer_wait_ticks = (unsigned long long)(atof(e) * 1e5); }
  { const char* e = getenv("CSSM_PEER_ALL_ROWS"); pf->peer_all_rows = e && e[0] == '1'; }
  { const char* e = getenv("CSSM_PEER_EAGER_ROWS"); if (e && atoll(e) >= 1) pf->peer_eager = atoll(e); }
  { const char* e = getenv("CSSM_WAVE_SUMS"); if (e) pf->opt_wave = atoi(e) == 2 ? 2 : (atoi(e) ? 1 : 0); }   // (A/B: the default of CSSM_OPT_WAVE_SUMS)
  { const char* e = getenv("CSSM_GRP_MIN_UNITS"); if (e && atoi(e) >= 1) pf->grp_min_units = (uint32_t)atoi(e); }   // (tests: small clouds through the group sums)
  HIP_TRY(hipDeviceGetAttribute(&pf->n_cus, hipDeviceAttributeMultiprocessorCount, pf->device));
  pf->stride = (size_t)((pf->n + CSSM_TILE - 1) / CSSM_TILE) * CSSM_TILE;   // rows start 16-B aligned
  pf->ntiles = (uint32_t)((pf->n + CSSM_TILE - 1) / CSSM_TILE);
  pf->sup = (pf->ntiles + 1023u) / 1024u;
  // Single GPU, clouds beyond 2^22 particles: units of 4 tiles while that gives at most 4096 of them (2^23: 2048 units, 2^24: 4096), larger
  // units beyond -- instead of 1024 units of ever more tiles (16 at 2^24: 1025 k_offspring blocks walking 16 tiles each through three
  // barriers per tile, 1024 k_propagate blocks on 1536 resident slots at d = 1).  k_offspring finds its prefix through the sums of up
  // to 64 groups of 64 units (Scalars::grp, layout 2).  A shard keeps at most 1024 units (its exchange kernels' geometry).
  // CSSM_UNIT_MAX_TILES (A/B): the tiles per unit to aim for (default 4; 0: the old rule).
  if (!pf->sharded && pf->sup > 4u) {
    uint32_t want = 4u;
    if (const char* e = getenv("CSSM_UNIT_MAX_TILES")) want = (uint32_t)atoi(e);
    if (want >= 1u) {
      const uint32_t least = (pf->ntiles + (uint32_t)(CSSM_GRP_MAX * 64) - 1u) / (uint32_t)(CSSM_GRP_MAX * 64);   // at most 4096 units
      const uint32_t sup2 = want > least ? want : least;
      if (sup2 < pf->sup) pf->sup = sup2;
    }
  }
  pf->nunits = (pf->ntiles + pf->sup - 1) / pf->sup;
  {   // k_propagate: a block owns unit/split particles, a multiple of its CSSM_BLOCK * IT particles per iteration
    // (the kernel pipelines its tiles through LDS and wants several of them: one block per unit)
    // Clouds below 2^20 particles on one GPU: one tile of the kernel per block.  Up to ~2^18 particles every SIMD holds at
    // most one or two waves and a kernel's duration is the length of ONE wave's dependent instruction stream (~3 ns per
    // instruction, tools/launch_floor.hip; the launch itself is 3.1 us): one pair of particles per thread instead of two, in the
    // single-tile instantiation k_propagate_self<..., ONE> that requests everything position-dependent in its first round of
    // loads and draws its normals while the gathered rows travel.  Per observation, bench model, same process (tools/ab_fine.py):
    // 17.1 -> 12.6 us at N = 100 000, 21.9 -> 19.3 at 2^19, 28.1 -> 25.8 at 3 * 2^18.  From 2^20 particles on: whole units per
    // block (launch_propagate).
    pf->split = auto_split(pf);
  }
  const size_t nsums = (size_t)(pf->ntiles > 4 * pf->nunits ? pf->ntiles : 4 * pf->nunits);   // (up to four sub-units per unit)
  const size_t row = pf->stride * 8;
  for (int b = 0; b < 2; ++b) {
    if (hipMalloc(&pf->state[b], row * pf->d + 64) != hipSuccess)   // + spare bytes: k_propagate fetches 16 bytes per element
      return fail(CSSM_ENOMEM, "hipMalloc of %zu bytes failed", row * pf->d);
    HIP_TRY(hipMemsetAsync(pf->state[b], 0, row * pf->d, pf->stream));
  }
  if (hipMalloc(&pf->logw, row) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc logw");
  if (hipMalloc(&pf->endslot, pf->stride * 4) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc endslot");
  if (hipMalloc(&pf->anc, pf->stride * 4) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc anc");
  HIP_TRY(hipMalloc(&pf->tileS, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMalloc(&pf->tileS2, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMalloc(&pf->tileP, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMalloc(&pf->unitPre, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMalloc(&pf->tileW, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMemsetAsync(pf->tileW, 0, nsums * sizeof(cssm_u128), pf->stream));
  pf->s2_stride = (uint32_t)nsums;
  HIP_TRY(hipMalloc(&pf->s2buf, 2 * nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMemsetAsync(pf->s2buf, 0, 2 * nsums * sizeof(cssm_u128), pf->stream));
  HIP_TRY(hipHostMalloc((void**)&pf->h_sc, sizeof(Scalars), hipHostMallocMapped));
  memset(pf->h_sc, 0, sizeof(Scalars));
  HIP_TRY(hipHostGetDevicePointer((void**)&pf->hd_sc, pf->h_sc, 0));
  HIP_TRY(hipHostMalloc((void**)&pf->h_done, 64, hipHostMallocMapped));
  memset(pf->h_done, 0, 64);
  HIP_TRY(hipHostGetDevicePointer((void**)&pf->hd_done, pf->h_done, 0));
  // sub-unit entries past the last k_propagate block are never written and must read as zero sums
  HIP_TRY(hipMemsetAsync(pf->tileS, 0, nsums * sizeof(cssm_u128), pf->stream));
  HIP_TRY(hipMemsetAsync(pf->tileS2, 0, nsums * sizeof(cssm_u128), pf->stream));
  HIP_TRY(hipMemsetAsync(pf->tileP, 0, nsums * sizeof(cssm_u128), pf->stream));
  HIP_TRY(hipMalloc(&pf->sc, sizeof(Scalars)));
  HIP_TRY(hipMemsetAsync(pf->sc, 0, sizeof(Scalars), pf->stream));
  HIP_TRY(hipMalloc(&pf->d_m0, CSSM_MAX_DIM * 8));
  HIP_TRY(hipMalloc(&pf->d_sd0, CSSM_MAX_DIM * 8));
  HIP_TRY(hipMalloc(&pf->d_bounds, 64 * 8));
  HIP_TRY(hipMalloc(&pf->d_xch, 136 * 8));
  HIP_TRY(hipMemsetAsync(pf->d_xch, 0, 136 * 8, pf->stream));
  HIP_TRY(hipMemsetAsync(pf->anc, 0, pf->stride * 4, pf->stream));   // always addressable, also before the first resampling
  HIP_TRY(hipMalloc(&pf->d_logtab, sizeof(CSSM_TAB)));
  HIP_TRY(hipMemcpyAsync(pf->d_logtab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice, pf->stream));
  return upload_init_params(pf);
}

static int create_common(const cssm_model_desc* desc, uint64_t n_global, uint64_t first, uint64_t n_local,
                         uint64_t seed, int device, void* stream, bool sharded, cssm_pf** out) {
  if (!out) return fail(CSSM_EINVAL_ARG, "out is null");
  *out = nullptr;
  if (n_global < 1 || n_global > 0xffff0000ull) return fail(CSSM_EINVAL_ARG, "n_particles must be in [1, 2^32 - 2^16]");
  if (n_local < 1 || first + n_local > n_global) return fail(CSSM_ESHARD, "shard [%llu, +%llu) outside [0, %llu)",
                                                              (unsigned long long)first, (unsigned long long)n_local, (unsigned long long)n_global);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(CSSM_EINVAL_ARG, "device %d out of range (%d devices)", device, ndev);
  cssm_pf* pf = new cssm_pf();
  pf->device = device;
  pf->n_global = n_global; pf->first = first; pf->n = n_local; pf->seed = seed; pf->sharded = sharded;
  if (sharded || stream != nullptr) { pf->stream = (hipStream_t)stream; pf->own_stream = false; }
  pf->opt_fused = 1;   // two launches per observation at every size (measured with the slim single-GPU kernels: 18.7 vs 20.0 us at
                            // N = 100 000, 34.0 vs 35.4 at 2^20, 336 vs 356 at 2^24); an outlying observation is redone in place
  int rc = cssm_build_model(pf, desc, false);
  if (rc == CSSM_OK) rc = alloc_handle(pf);
  if (rc != CSSM_OK) { const std::string keep = cssm_last_error(); cssm_pf_destroy(pf); return fail(rc, "%s", keep.c_str()); }
  *out = pf;
  return CSSM_OK;
}

// a single-GPU handle on a stream of the caller's (the chains of a batch share one: cssm_batch.hip)
int cssm_pf_create_on_stream(const cssm_model_desc* desc, uint64_t n_particles, uint64_t seed, int device, hipStream_t stream, cssm_pf** out) {
  return create_common(desc, n_particles, 0, n_particles, seed, device, stream, false, out);
}

extern "C" int cssm_pf_create(const cssm_model_desc* desc, uint64_t n_particles, uint64_t seed, int device, cssm_pf** out) {
  return create_common(desc, n_particles, 0, n_particles, seed, device, nullptr, false, out);
}

extern "C" int cssm_pf_create_shard(const cssm_model_desc* desc, uint64_t n_global, uint64_t first, uint64_t n_local,
                                    uint64_t seed, int device, void* hip_stream, cssm_pf** out) {
  return create_common(desc, n_global, first, n_local, seed, device, hip_stream, true, out);
}

extern "C" void cssm_pf_destroy(cssm_pf* pf) {
  if (!pf) return;
  (void)hipSetDevice(pf->device);
  if (pf->stream) (void)hipStreamSynchronize(pf->stream);
  cssm_peer_free(pf);
  void* ptrs[] = {pf->sm_keys, pf->sm_partial, pf->sm_st, pf->sm_rec, pf->s2buf, pf->fineS, pf->fineS2, pf->state[0], pf->state[1], pf->logw, pf->endslot, pf->anc, pf->tileS, pf->tileS2, pf->tileP, pf->unitPre, pf->tileW, pf->sc,
                  pf->d_m0, pf->d_sd0, pf->d_logtab, pf->d_fsub, pf->cum, pf->d_recs, pf->d_ll_t, pf->d_ess_t, pf->d_path, pf->cand, pf->cand_end, pf->cand_idx, pf->d_bounds, pf->d_xch, pf->d_need};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (pf->h_recs) (void)hipHostFree(pf->h_recs);
  if (pf->h_sc) (void)hipHostFree(pf->h_sc);
  if (pf->h_done) (void)hipHostFree(pf->h_done);
  if (pf->h_ll_t) (void)hipHostFree(pf->h_ll_t);
  if (pf->h_ess_t) (void)hipHostFree(pf->h_ess_t);
  for (hipEvent_t e : pf->prof_ev) (void)hipEventDestroy(e);
  if (pf->ev0) (void)hipEventDestroy(pf->ev0);
  if (pf->ev1) (void)hipEventDestroy(pf->ev1);
  if (pf->own_stream && pf->stream) (void)hipStreamDestroy(pf->stream);
  delete pf;
}

extern "C" int cssm_pf_set_params(cssm_pf* pf, const cssm_model_desc* desc) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = cssm_build_model(pf, desc, true);
  if (rc) return rc;
  return upload_init_params(pf);
}

extern "C" int cssm_pf_reseed(cssm_pf* pf, uint64_t seed) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  pf->seed = seed;
  return CSSM_OK;
}

extern "C" uint64_t cssm_pf_run_key(uint64_t seed, uint64_t run) { return cssm_derive_key(seed, run); }

extern "C" uint64_t cssm_pf_num_particles(const cssm_pf* pf) { return pf ? pf->n : 0; }
extern "C" int32_t cssm_pf_dim(const cssm_pf* pf) { return pf ? pf->d : 0; }
// ------------------------------------------------------------------------------------ launches

// ll = 0.0, ess = N: PfState(t0, None, state, 0.0, particles), model/ParticleFilter.scala:107
static int reset_scalars(cssm_pf* pf) {
  Scalars h;
  memset(&h, 0, sizeof h);
  h.ess = (int32_t)(pf->n_global < 2147483647ull ? pf->n_global : 2147483647ull);
  h.fail_step = 0xffffffffu;
  h.next_ref = cssm_nan();      // (LGCP: no weighted observation yet whose max could predict a level)
  h.peer_wait_ticks = pf->peer_wait_ticks;
  // the slot and group-sum sets (120 KB since round 5: a PMMH chain initialises a filter per iteration) are cleared on the device; only
  // the scalars behind them travel (pageable source: the copy is staged before the call returns, so a stack object is safe)
  HIP_TRY(hipMemsetAsync(pf->sc, 0, offsetof(Scalars, err), pf->stream));
  HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(pf->sc) + CSSM_SC_TAIL_OFF, reinterpret_cast<const char*>(&h) + CSSM_SC_TAIL_OFF, sizeof(Scalars) - CSSM_SC_TAIL_OFF,
                         hipMemcpyHostToDevice, pf->stream));
  return CSSM_OK;
}

// host-side state of a handle whose cloud has just been drawn (or replicated) into state[0]: what cssm_launch_init and
// cssm_pf_init_from share (reset_scalars cleared the device side, Scalars::pend included)
static void fresh_host_state(cssm_pf* pf, double t0) {
  pf->cur = 0; pf->src = pf->state[0]; pf->src_stride = pf->stride; pf->src2 = nullptr; pf->anc_valid = false;
  pf->t = t0; pf->step = 0; pf->initialised = true; pf->wparity = 0;
  pf->wmode = false; pf->sums_ready = false; pf->last_optimistic = false; pf->last_grp = false; pf->have_level = false;
  pf->ess_host = (int32_t)(pf->n_global < 2147483647ull ? pf->n_global : 2147483647ull);
}

int cssm_launch_init(cssm_pf* pf, double t0) {
  HIP_TRY(hipSetDevice(pf->device));
  const int grid = grid_for(pf->n, CSSM_BLOCK, kGridCap);
  DISPATCH_D(pf->d, k_init<D><<<dim3(grid), dim3(CSSM_BLOCK), 0, pf->stream>>>(pf->state[0], pf->stride, pf->n, pf->first,
                                                                              pf->seed, pf->d_m0, pf->d_sd0, pf->d_logtab));
  HIP_TRY(hipGetLastError());
  int rc = reset_scalars(pf);
  if (rc) return rc;
  fresh_host_state(pf, t0);
  return CSSM_OK;
}

// propagate + weight of one datum (record already on the device)

// whether launch_propagate will use the kernels that also form the sums (and can record sampleOne's pick on the way)
// (LGCP, contract v8: once a weighted observation has run natively its max predicts the next one's level -- Scalars::next_ref --
//  and the sums are formed inside the propagate like everybody else's; the first observation's level is its max)
static bool uses_sums_kernel(const cssm_pf* pf) {
  return pf->opt_fused && !pf->safe_sums && (pf->obs_kind != CSSM_OBS_LGCP || pf->have_level) && pf->resampler != CSSM_RESAMPLE_MULTINOMIAL;
}
// ... ever, in the series that is being enqueued (whether an observation can put it on hold)
static bool may_use_sums_kernel(const cssm_pf* pf) {
  return pf->opt_fused && !pf->safe_sums && pf->resampler != CSSM_RESAMPLE_MULTINOMIAL;
}

// Records [first, first + count) of the handle's host buffer -> the device.  chain (the records CONTINUE a running filter): where
// the level is predicted from the observation before (LGCP: StepRec::predict) the first record's comes from Scalars::next_ref,
// on the stream; later records of the call get theirs from the kernels that publish their predecessors (publish_next_level).
// The observation records travel by a KERNEL that reads the pinned host buffer (one block per record), not by hipMemcpyAsync: a
// 14 KB copy cost 7-11 us of host time and left the queue idle for 9 us before the first propagate (rocprofv3 timeline of 20-step
// legs: copy 0 .. 3, first kernel at 11.6 us); a launch costs 3.5 us of host time and the records' 89 words cross PCIe in one round trip.
// CSSM_UPLOAD_MEMCPY=1 keeps the copy (A/B).
__global__ __launch_bounds__(128) void k_fetch_recs(const StepRec* __restrict__ host, StepRec* __restrict__ dev, Scalars* __restrict__ sc) {
  static_assert(sizeof(StepRec) % 8 == 0, "records are copied in 8-byte words");
  if (sc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) sc->t_first = __builtin_amdgcn_s_memrealtime();   // (the call's first kernel: Scalars::t_first)
  const unsigned long long* s = reinterpret_cast<const unsigned long long*>(host + blockIdx.x);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dev + blockIdx.x);
  for (uint32_t i = threadIdx.x; i < sizeof(StepRec) / 8; i += 128) d[i] = s[i];
}
// ONE record (a streaming step; the first observation of a continued batch call): it travels in the kernel's ARGUMENTS -- no read
// across PCIe at all (the argument segment lives in device memory), the block copies its own kernarg bytes to the record's slot
__global__ __launch_bounds__(128) void k_put_rec(StepRec r, StepRec* __restrict__ dev, Scalars* __restrict__ sc) {
  static_assert(offsetof(StepRec, step) < sizeof(StepRec), "");
  if (sc != nullptr && threadIdx.x == 0) sc->t_first = __builtin_amdgcn_s_memrealtime();   // (the call's first kernel: Scalars::t_first)
  const unsigned long long* s = (const unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();   // (r is the first argument: offset 0)
  unsigned long long* d = reinterpret_cast<unsigned long long*>(dev);
  if (threadIdx.x < sizeof(StepRec) / 8) d[threadIdx.x] = s[threadIdx.x];
  (void)r;
}
int cssm_upload_recs(cssm_pf* pf, size_t first, size_t count, bool chain) {
  Scalars* stamp = first == 0 ? pf->sc : nullptr;   // (record 0 of a call travels with the call's first kernel)
  static const bool by_copy = getenv("CSSM_UPLOAD_MEMCPY") != nullptr;
  static const bool no_kernarg = getenv("CSSM_UPLOAD_NO_KERNARG") != nullptr;
  if (count == 0) return CSSM_OK;
  if (count == 1 && !by_copy && !no_kernarg) {
    static_assert(sizeof(StepRec) / 8 <= 128 && sizeof(StepRec) <= 3072, "one record fits a kernel's argument segment and one block copies it");
    hipLaunchKernelGGL(k_put_rec, dim3(1), dim3(128), 0, pf->stream, pf->h_recs[first], pf->d_recs + first, stamp);
    HIP_TRY(hipGetLastError());
  } else if (by_copy || !pf->h_recs_dev || count > 0x7fffffffu) {
    HIP_TRY(hipMemcpyAsync(pf->d_recs + first, pf->h_recs + first, count * sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  } else {
    hipLaunchKernelGGL(k_fetch_recs, dim3((uint32_t)count), dim3(128), 0, pf->stream, (const StepRec*)(pf->h_recs_dev + first), pf->d_recs + first, stamp);
    HIP_TRY(hipGetLastError());
  }
  if (chain && pf->obs_kind == CSSM_OBS_LGCP) {
    hipLaunchKernelGGL(k_chain_level, dim3(1), dim3(1), 0, pf->stream, pf->d_recs + first, (const Scalars*)pf->sc);
    HIP_TRY(hipGetLastError());
  }
  return CSSM_OK;
}

// Launch geometry of the sums kernels on clouds of 2^20 particles and more (one GPU; whole units of 1024 * sup particles).
// In-process A/B (tools/ab_opt.py option 6, tools/ab_fine.py; per observation; boxes of this pool differ by +-5 %, only runs of
// one process compare):
//   LOOP    one block per unit running the single-tile-style kernel tile after tile (k_propagate_self<..., ONE>: no software-
//           pipeline state, no spill, 5-6 waves per SIMD; co-resident waves cover the round trips).  While a block has at most
//           CSSM_LOOP_MAX_TILES tiles it beats everything else: d = 3: 31.0 vs 33.2 (half tiles) / 32.5 (pipelined) us at 2^20,
//           50.7 vs 53.3 at 2^21, 90.0 vs 93.0 at 2^22, a tie at 2^23, 376 vs 368 at 2^24; d = 1: 23.6 vs 26.9 at 2^20, a tie at 2^22,
//           252 vs 241 at 2^24; d = 4: 108 vs 120 at 2^22; d = 9: 57.6 vs 63.4 at 2^20, 105 vs 114 at 2^21; d = 16 at 2^22 (16 tiles): 329 vs 313.
//   FINE    one tile per block + k_reduce_units (~5 us): larger clouds of d >= 4 (d = 9: 194 vs 203 us pipelined at 2^22; d = 16: 287 vs 343)
//   PIPE    one block per unit, the software-pipelined kernel: larger clouds of d <= 3 (a tie with FINE at d = 3, a few per cent better at d <= 2)
#ifndef CSSM_FINE_MIN_D
#define CSSM_FINE_MIN_D 4
#endif
#ifndef CSSM_LOOP_MAX_TILES
#define CSSM_LOOP_MAX_TILES 8
#endif
enum { GEO_PIPE = 0, GEO_LOOP = 1, GEO_FINE = 2 };
static int large_geometry(const cssm_pf* pf) {
  if (pf->sharded || pf->split != 1 || pf->first != 0 || pf->n != pf->n_global || pf->resampler == CSSM_RESAMPLE_MULTINOMIAL) return GEO_PIPE;
  if (pf->opt_whole == 1) return GEO_LOOP;
  if (pf->opt_whole == 2) return GEO_PIPE;
  if (pf->opt_whole == 3) return GEO_FINE;
  const uint32_t tiles = pf->sup * (uint32_t)CSSM_TILE / (uint32_t)(CSSM_BLOCK * cssm_prop_items(pf->d));
  if (tiles <= CSSM_LOOP_MAX_TILES) return GEO_LOOP;
  return pf->d >= CSSM_FINE_MIN_D ? GEO_FINE : GEO_PIPE;
}

void cssm_batch_fresh(cssm_pf* pf, double t0) { fresh_host_state(pf, t0); pf->gen++; pf->done_seq++; }
int cssm_batch_ok(const cssm_pf* pf) {
  return pf->opt_fused && !pf->safe_sums && pf->obs_kind != CSSM_OBS_LGCP && pf->resampler == CSSM_RESAMPLE_SYSTEMATIC && !pf->sharded &&
         large_geometry(pf) != GEO_FINE;
}

int cssm_launch_propagate(cssm_pf* pf, const StepRec* d_rec, double* pick_out, uint32_t pick_slot, const CssmBatchLaunch* batch) {
  // The observation's index travels as a kernel argument (the slim kernels use it before any load lands): every record a
  // propagate is launched on lives in the handle's record buffer, and the HOST copy of that record -- h_recs mirrors d_recs from
  // the moment a record is built until the launch that consumes it: every caller builds the record, enqueues its upload and
  // launches, in that order, and never rewrites a slot with launches on it still to come -- supplies the index.  Checked before
  // anything of the handle changes: a refused launch leaves no trace.
  if (!pf->h_recs || d_rec < pf->d_recs || d_rec >= pf->d_recs + std::min(pf->h_recs_cap, pf->recs_cap))
    return fail(CSSM_ESTATE, "propagate launched on a record outside the handle's record buffer");
  const uint32_t rec_step = pf->h_recs[d_rec - pf->d_recs].step;
  // one block per sub-unit: contiguous ranges, so that (with do_sums) the block's fixed-point sums are the
  // sub-unit sums k_offspring scans
  uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / pf->split;
  double* dst = pf->state[pf->cur ^ 1];
  const uint32_t* anc = pf->anc_valid ? pf->anc : nullptr;
  const int do_sums = uses_sums_kernel(pf) ? 1 : 0;
  pf->last_optimistic = do_sums != 0;
  const int geo = do_sums ? large_geometry(pf) : GEO_PIPE;
  const bool fine = geo == GEO_FINE;
  const uint64_t unit_particles = (uint64_t)pf->sup * CSSM_TILE;
  if (fine) chunk = (uint64_t)CSSM_BLOCK * cssm_prop_items(pf->d);   // (divides the unit: 1024 * sup)
  const int grid = (int)((pf->n + chunk - 1) / chunk);
  if (fine && pf->fine_cap < (size_t)grid) {
    HIP_TRY(hipStreamSynchronize(pf->stream));
    if (pf->fineS) (void)hipFree(pf->fineS);
    if (pf->fineS2) (void)hipFree(pf->fineS2);
    pf->fineS = pf->fineS2 = nullptr; pf->fine_cap = 0;
    if (hipMalloc(&pf->fineS, (size_t)grid * sizeof(cssm_u128)) != hipSuccess || hipMalloc(&pf->fineS2, (size_t)grid * sizeof(cssm_u128)) != hipSuccess)
      return fail(CSSM_ENOMEM, "hipMalloc of the per-block sums (%d blocks)", grid);
    pf->fine_cap = (size_t)grid;
  }
  prof_begin(pf, CSSM_K_PROPAGATE);
  PropLaunch a;
  a.chains = nullptr; a.nchains = 0; a.cur = pf->cur; a.anc_valid = pf->anc_valid ? 1 : 0; a.want_pick = 0; a.rec_idx = 0;
  if (batch) { a.chains = batch->chains; a.nchains = batch->nchains; a.want_pick = batch->want_pick; a.rec_idx = batch->rec_idx; }
  a.grid = grid; a.stream = pf->stream;
  a.lgcp = pf->obs_kind == CSSM_OBS_LGCP;
  a.sharded = pf->sharded ? 1 : 0;
  a.obs = (pf->obs_kind == CSSM_OBS_POISSON || pf->obs_kind == CSSM_OBS_GAUSSIAN) ? pf->obs_kind : -1;
  a.sums = do_sums;
  a.src = pf->src; a.src_stride = pf->src_stride; a.anc = anc; a.dst = dst; a.dst_stride = pf->stride; a.logw = pf->logw;
  a.n = pf->n; a.gid0 = pf->first; a.seed = pf->seed; a.rec = d_rec; a.mk = pf->mk; a.sc = pf->sc;
  a.slot_set = pf->sharded ? 0 : pf->wparity;
  // one fused-sums block per unit of a single-GPU cloud: the blocks also accumulate the sums of groups of 32 units (Scalars::grp,
  // bit 8 of the set argument of k_propagate_self and k_offspring_self), which k_offspring then reads instead of every unit sum
  // (a shard: the same, in Scalars::grp and grp2 -- both sums travel in the exchange's headers --, the set rotating with the handle's
  //  weighted observations while its max slots stay in set 0; the exchange kernels read 32 group sums instead of every unit sum)
  // layout 1: at most 32 groups of 32 units; layout 2 (single GPU, not the batched chains): at most 64 groups of 64 units
  const bool big = pf->nunits > (uint32_t)(CSSM_GRP_SMALL * CSSM_GRP_UNITS);
  pf->last_grp = do_sums && !fine && chunk * pf->split == unit_particles && (pf->sharded ? pf->split <= 4u : (pf->split == 1 && pf->first == 0 && pf->n == pf->n_global)) &&
                 pf->nunits >= pf->grp_min_units && (big ? (!pf->sharded && batch == nullptr && pf->nunits <= (uint32_t)(CSSM_GRP_MAX * 64)) : true) &&
                 unit_particles <= CSSM_GRP_MAX_UNIT && pf->opt_grp;
  pf->grp_layout = pf->last_grp ? (big ? 2 : 1) : 0;
  const bool want_grp = pf->last_grp;
  // bits 11-12: log2(blocks per group / 32) -- the blocks per unit of a shard's split launch, or the 64-unit groups of layout 2
  if (want_grp) a.slot_set |= 0x100 | (pf->wparity << 9) | ((big ? 1 : (pf->split == 4u ? 2 : (pf->split == 2u ? 1 : 0))) << 11);
  // ... and, on the single GPU's tile-after-tile launch behind systematic resampling, the exact sums of the waves' quarter units (bit 13;
  // the array travels in the slot of the sums of squares, which these kernels leave to k_offspring): k_offspring_wave then needs neither
  // a conversion nor a 128-bit scan per weight (CSSM_OPT_WAVE_SUMS = 0: k_offspring_self as before)
  const bool want_ws = want_grp && geo == GEO_LOOP && !pf->sharded && batch == nullptr && pf->resampler == CSSM_RESAMPLE_SYSTEMATIC &&
                       pf->obs_kind != CSSM_OBS_LGCP && pick_out == nullptr && 4 * (size_t)pf->nunits <= (size_t)pf->s2_stride &&
                       // (where it pays: units of several tiles -- clouds beyond 2^20 particles: at one tile per unit the resampling kernel gains
                       //  nothing back -- and at most two latent components: a block on wave ranges streams 4 x (d rows + weights) ranges instead
                       //  of d + 1, and from d = 3 on the propagate loses to that what the resampling kernel gains (same-box A/B of round 6: the
                       //  step -1.2 % on slower boxes, +0-5 % on the fastest, where the block-wide propagate runs at 0.68 of the HBM peak; d = 1:
                       //  -4.8 % .. 0).  CSSM_OPT_WAVE_SUMS = 2 forces it wherever the geometry allows.)
                       (pf->opt_wave == 2 || (pf->opt_wave != 0 && pf->sup >= 2u && pf->d <= 2));
  if (want_ws) a.slot_set |= 0x2000;
  a.src2 = anc ? pf->src2 : nullptr; a.src2_stride = pf->src2_stride; a.n_split = pf->n_split; a.logtab = pf->d_logtab;
  a.chunk = chunk; a.do_sums = do_sums; a.subS = fine ? pf->fineS : pf->tileS; a.subS2 = want_ws ? pf->tileW : (fine ? pf->fineS2 : pf->tileS2);
  a.pick_out = pick_out; a.pick_slot = pick_slot;
  a.step = rec_step;
  a.fsub = pf->lgcp_tdep ? pf->d_fsub : nullptr;
  a.one = (chunk == (uint64_t)CSSM_BLOCK * cssm_prop_items(pf->d)) ? 1 : (geo == GEO_LOOP ? 2 : 0);
  a.specialise = pf->opt_spec; a.obs_kind = pf->obs_kind;
  // sharded handle on the single-collective exchange (received rows read in place: src2_stride == 0) or before its first exchange:
  // slim launch, tile after tile while a unit has at most CSSM_LOOP_MAX_TILES tiles; whole pairs per thread (d <= 8) need an even first id
  a.shard_slim = pf->sharded && do_sums && !a.lgcp && (a.src2 == nullptr || a.src2_stride == 0) && a.fsub == nullptr && pick_out == nullptr &&
                 ((pf->first & 1ull) == 0ull || cssm_prop_items(pf->d) == 1) && (a.slot_set & 0xff) == 0;
  if (a.shard_slim && pf->sup * (uint32_t)CSSM_TILE / (uint32_t)(CSSM_BLOCK * cssm_prop_items(pf->d)) <= CSSM_LOOP_MAX_TILES) a.one = 2;
  int launched = 0;   // CSSM_PROP_LAUNCHED_* of the kernel the dispatcher chose
  switch (pf->d) {
#define CSSM_CASE_PROP(D) case D: launched = cssm_prop_launch_d##D(a); break;
    CSSM_CASE_PROP(1) CSSM_CASE_PROP(2) CSSM_CASE_PROP(3) CSSM_CASE_PROP(4) CSSM_CASE_PROP(5) CSSM_CASE_PROP(6) CSSM_CASE_PROP(7) CSSM_CASE_PROP(8)
    CSSM_CASE_PROP(9) CSSM_CASE_PROP(10) CSSM_CASE_PROP(11) CSSM_CASE_PROP(12) CSSM_CASE_PROP(13) CSSM_CASE_PROP(14) CSSM_CASE_PROP(15)
    default: launched = cssm_prop_launch_d16(a); break;
#undef CSSM_CASE_PROP
  }
  // k_offspring reads the group sums only where the kernel that ran accumulated them (the unit sums exist either way)
  pf->last_grp = want_grp && (launched & CSSM_PROP_LAUNCHED_GRP) != 0;
  pf->last_ws = pf->last_grp && want_ws && (launched & CSSM_PROP_LAUNCHED_WS) != 0;
  prof_end(pf);
  if (fine) {   // the blocks' sums -> the units' (the kernel itself skips unweighted observations and series on hold)
    prof_begin(pf, CSSM_K_REDUCE);
    hipLaunchKernelGGL(k_reduce_units, dim3((pf->nunits + CSSM_BLOCK / 64 - 1) / (CSSM_BLOCK / 64)), dim3(CSSM_BLOCK), 0, pf->stream,
                       (const cssm_u128*)pf->fineS, (const cssm_u128*)nullptr, (uint32_t)grid, (uint32_t)(unit_particles / chunk), pf->nunits,
                       pf->tileS, pf->tileS2, (const Scalars*)pf->sc, d_rec);   // (single GPU: the squares are k_offspring's)
    prof_end(pf);
  }
  HIP_TRY(hipGetLastError());
  pf->cur ^= 1;
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->anc_valid = false; pf->src2 = nullptr;
  return CSSM_OK;
}

// sums (unless k_propagate formed them) -> end slots -> ancestors, single GPU.  After a k_propagate<SUMS> the buffer holds the
// weights themselves (raw = 2) and k_offspring forms the sum of squares on the way: that observation's ESS stays pending until
// the next weighted observation's publisher block, or the host at the end of the call, totals the blocks' partials.
// (allow_grp = false: a REDONE observation -- the set of group sums it would add to still holds the sums of its first attempt)
static int launch_resample(cssm_pf* pf, const StepRec* d_rec, double* ll_t = nullptr, int32_t* ess_t = nullptr, uint32_t rec_idx = 0, bool allow_grp = true) {
#ifdef CSSM_OFF_STAMPS
  if (!pf->cum) { HIP_TRY(hipMalloc(&pf->cum, pf->stride * 8 + (1u << 20))); HIP_TRY(hipMemsetAsync(pf->cum, 0, 1u << 20, pf->stream)); }
#endif
  if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL && !pf->cum) {
    if (hipMalloc(&pf->cum, pf->stride * 8) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc cumulative weights");
  }
  const bool optimistic = pf->last_optimistic;
  const int tgrid = (int)pf->nunits;
  const int split = optimistic ? (int)pf->split : 1;
  // the sums as a pass of their own on a cloud of many units (systematic resampling, one GPU): k_tile_sums adds the units' sums to their
  // groups' too, and k_offspring_self<..., 0, GRP> finds its prefix through those instead of every block reading every unit sum (at
  // 4096 units: 64 KiB per block, 100 us per launch at N = 2^24 against 67)
  int lean_layout = 0;
  if (!optimistic && allow_grp && !pf->sharded && pf->opt_grp && pf->resampler == CSSM_RESAMPLE_SYSTEMATIC && pf->first == 0 && pf->n == pf->n_global &&
      pf->nunits >= pf->grp_min_units && pf->nunits <= (uint32_t)(CSSM_GRP_MAX * 64) && (uint64_t)pf->sup * CSSM_TILE <= CSSM_GRP_MAX_UNIT)
    lean_layout = pf->nunits > (uint32_t)(CSSM_GRP_SMALL * CSSM_GRP_UNITS) ? 2 : 1;
  if (!optimistic) {
    prof_begin(pf, CSSM_K_TILE_SUMS);
    hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc, pf->tileS, pf->tileS2, pf->ntiles,
                       pf->sup, pf->nunits, 0, pf->wparity, (const double*)nullptr, pf->d_logtab, d_rec, 0u, (const unsigned long long*)nullptr, 0,
                       lean_layout ? (0x100 | (pf->wparity << 9) | ((lean_layout == 2 ? 1 : 0) << 11)) : 0);
    prof_end(pf);
  }
  const int s2_par = optimistic ? pf->s2_par : -1;
  prof_begin(pf, CSSM_K_OFFSPRING);   // unit prefix, ll (ess), end slots and their expansion to ancestors in one kernel (one block per unit + the publisher)
#define OFF_ARGS pf->logw, pf->n, pf->sc, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, d_rec, pf->anc, pf->ntiles, pf->sup, pf->nunits, \
                 pf->wparity, ll_t, ess_t, rec_idx, pf->opt_exact, split, pf->seed, pf->cum, pf->s2buf, pf->s2_stride, s2_par, pf->gen
  const int ogrid = tgrid + 1;   // one block per unit + the publisher
#define OFF_WAVE_ARGS pf->logw, pf->n, pf->sc, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileW, d_rec, pf->anc, pf->sup, pf->nunits, pf->wparity, \
                      ll_t, ess_t, rec_idx, pf->opt_exact, pf->s2buf, pf->s2_stride, s2_par, pf->gen
  static const bool no_offw = getenv("CSSM_NO_OFFW") != nullptr;   // (A/B: the wave-range propagate in front of k_offspring_self)
#define OFF_GO(RS) do { if (optimistic && pf->last_grp && pf->last_ws && !no_offw && RS == CSSM_RESAMPLE_SYSTEMATIC) { \
                          if (pf->grp_layout == 2) hipLaunchKernelGGL((k_offspring_wave<2>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_WAVE_ARGS); \
                          else hipLaunchKernelGGL((k_offspring_wave<1>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_WAVE_ARGS); \
                        } else if (optimistic && pf->last_grp && pf->grp_layout == 2 && RS == CSSM_RESAMPLE_SYSTEMATIC) \
                          hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_SYSTEMATIC, 2, 2>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); \
                        else if (optimistic && pf->last_grp && RS == CSSM_RESAMPLE_SYSTEMATIC) \
                          hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_SYSTEMATIC, 2, 1>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); \
                        else if (optimistic) hipLaunchKernelGGL((k_offspring_self<RS, 2>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); \
                        else if (lean_layout == 2 && RS == CSSM_RESAMPLE_SYSTEMATIC) \
                          hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_SYSTEMATIC, 0, 2>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); \
                        else if (lean_layout == 1 && RS == CSSM_RESAMPLE_SYSTEMATIC) \
                          hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_SYSTEMATIC, 0, 1>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); \
                        else hipLaunchKernelGGL((k_offspring_self<RS, 0>), dim3(ogrid), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS); } while (0)
  if (pf->resampler == CSSM_RESAMPLE_STRATIFIED) OFF_GO(CSSM_RESAMPLE_STRATIFIED);
  else if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL) OFF_GO(CSSM_RESAMPLE_MULTINOMIAL);
  else OFF_GO(CSSM_RESAMPLE_SYSTEMATIC);
#undef OFF_GO
#undef OFF_ARGS
#undef OFF_WAVE_ARGS
  if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL)
    hipLaunchKernelGGL(k_multinomial, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, pf->cum, pf->n, pf->seed,
                       pf->h_step_for_resample, pf->anc);
  prof_end(pf);
  pf->wparity = (pf->wparity + 1) % CSSM_MAXSETS;
  HIP_TRY(hipGetLastError());
  pf->anc_valid = true;
  pf->wmode = optimistic;
  pf->have_level = true;        // (the publisher of this launch leaves the next observation's predicted level: LGCP)
  if (optimistic) pf->s2_par ^= 1;
  return CSSM_OK;
}
static int launch_step(cssm_pf* pf, const StepRec* d_rec, int weighted, uint32_t step_index, double* ll_t = nullptr, int32_t* ess_t = nullptr,
                       uint32_t rec_idx = 0, double* pick_out = nullptr, uint32_t pick_slot = 0) {
  pf->h_step_for_resample = step_index;
  int rc = cssm_launch_propagate(pf, d_rec, pick_out, pick_slot);
  if (rc) return rc;
  if (weighted) rc = launch_resample(pf, d_rec, ll_t, ess_t, rec_idx);
  else if (ll_t) hipLaunchKernelGGL(k_record, dim3(1), dim3(1), 0, pf->stream, pf->sc, ll_t, ess_t, rec_idx);
  return rc;
}

int cssm_check_device_err(cssm_pf* pf, const Scalars& h) {
  if (h.err & 1u) return fail(CSSM_ENONFINITE, "a log-weight is NaN (the reference's breeze distribution constructor would throw)");
  if (h.err & 2u) return fail(CSSM_ENONFINITE, "all particle weights are zero or the maximum log-weight is not finite");
  if (h.err & 16u) return fail(CSSM_ESHARD, "peer-written exchange: a rank's segment did not arrive within the wait bound");
  (void)pf;
  return CSSM_OK;
}
// The end of a call: k_finish forms an ESS that is still pending, and writes the scalars the host reads (and, with T > 0, the
// call's ll_t / ess_t) into host-mapped memory; the host synchronises the stream -- no device-to-host copy.  *pf->h_sc is valid
// from `err` on afterwards.
// poll = true (the batch drivers without a path copy, not while profiling): the host does not wait for the STREAM but for the word
// k_finish stores last (system-scope release behind its other stores into host-mapped memory) -- the runtime's wait returned
// ~10 us behind the kernel's end, a short continued leg (bench.py --steps 20) pays that once per 20 observations.  The stream is
// queried now and then: a faulted or vanished kernel must not leave the host spinning.
static int read_scalars(cssm_pf* pf, uint32_t T = 0, bool want_ll_t = false, bool want_ess_t = false, bool poll = false) {
  const uint32_t seq = ++pf->done_seq;
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(CSSM_BLOCK), 0, pf->stream, pf->sc, (const cssm_u128*)pf->s2buf, pf->s2_stride, pf->d_ll_t, pf->d_ess_t, T,
                     pf->gen, pf->hd_sc, want_ll_t ? pf->hd_ll_t : (double*)nullptr, want_ess_t ? pf->hd_ess_t : (int32_t*)nullptr,
                     pf->hd_done, seq);
  HIP_TRY(hipGetLastError());
  static const bool no_poll = getenv("CSSM_NO_POLL") != nullptr;
  if (poll && !no_poll && !pf->profile) {
    volatile uint32_t* flag = pf->h_done;
    for (uint64_t spins = 1;; ++spins) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
      if ((spins & 0x3fffu) == 0u) {             // every ~16 K polls (tens of microseconds): is the stream still alive?
        const hipError_t q = hipStreamQuery(pf->stream);
        if (q == hipSuccess) {                    // drained: the word must be there (else the kernel did not run to its end)
          if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
          return fail(CSSM_EHIP, "the stream drained without k_finish's completion word");
        }
        if (q != hipErrorNotReady) return fail(CSSM_EHIP, "HIP error while waiting for the series: %s", hipGetErrorString(q));
      }
    }
  } else {
    HIP_TRY(hipStreamSynchronize(pf->stream));
  }
  pf->ess_host = pf->h_sc->ess;
  return CSSM_OK;
}

// Host-side state of a single-GPU handle right before a propagate: what redoing that observation starts from.
struct PreState { int cur; const double* src; size_t src_stride; bool anc_valid; int wparity; };
static inline PreState pre_state(const cssm_pf* pf) { return {pf->cur, pf->src, pf->src_stride, pf->anc_valid, pf->wparity}; }
// An observation whose reference level the max ruled out (err bit 6: k_offspring put the series on hold AT it; its propagate
// is done, the cloud it read and the previous ancestors are untouched) is done AGAIN: the same propagate with log-weights
// stored (the weights relative to the level it first assumed are useless), sums relative to the max (k_tile_sums), resampling.
static int redo_observation(cssm_pf* pf, const PreState& q, const StepRec* d_rec, double* ll_t, int32_t* ess_t, uint32_t rec_idx) {
  pf->cur = q.cur; pf->src = q.src; pf->src_stride = q.src_stride; pf->src2 = nullptr; pf->anc_valid = q.anc_valid; pf->wparity = q.wparity;
  pf->safe_sums = true;
  int rc = cssm_launch_propagate(pf, d_rec, nullptr, 0);
  pf->safe_sums = false;
  if (rc) return rc;
  return launch_resample(pf, d_rec, ll_t, ess_t, rec_idx, /*allow_grp=*/false);
}

// ------------------------------------------------------------------------------------ streaming API

int cssm_ensure_recs(cssm_pf* pf, size_t T) {
  if (T < 1024) T = 1024;   // one allocation serves every ordinary series length
  if (pf->h_recs_cap < T) {
    if (pf->h_recs) (void)hipHostFree(pf->h_recs);
    pf->h_recs = nullptr;
    HIP_TRY(hipHostMalloc((void**)&pf->h_recs, T * sizeof(StepRec), hipHostMallocDefault));
    pf->h_recs_cap = T;
    pf->h_recs_dev = nullptr;   // (the device's address of the pinned buffer: k_fetch_recs reads it; without one the records are copied)
    if (hipHostGetDevicePointer((void**)&pf->h_recs_dev, pf->h_recs, 0) != hipSuccess) { (void)hipGetLastError(); pf->h_recs_dev = nullptr; }
  }
  if (pf->recs_cap < T) {
    if (pf->d_recs) (void)hipFree(pf->d_recs);
    if (pf->d_ll_t) (void)hipFree(pf->d_ll_t);
    if (pf->d_ess_t) (void)hipFree(pf->d_ess_t);
    if (pf->h_ll_t) (void)hipHostFree(pf->h_ll_t);
    if (pf->h_ess_t) (void)hipHostFree(pf->h_ess_t);
    pf->d_recs = nullptr; pf->d_ll_t = nullptr; pf->d_ess_t = nullptr; pf->h_ll_t = pf->hd_ll_t = nullptr; pf->h_ess_t = pf->hd_ess_t = nullptr;
    HIP_TRY(hipMalloc(&pf->d_recs, (T + 1) * sizeof(StepRec)));   // (+ 1: publish_next_level writes the level of the record BEHIND the one it publishes)
    HIP_TRY(hipMalloc(&pf->d_ll_t, T * 8));
    HIP_TRY(hipMalloc(&pf->d_ess_t, T * 4));
    HIP_TRY(hipHostMalloc((void**)&pf->h_ll_t, T * 8, hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&pf->h_ess_t, T * 4, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&pf->hd_ll_t, pf->h_ll_t, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&pf->hd_ess_t, pf->h_ess_t, 0));
    pf->recs_cap = T;
  }
  return CSSM_OK;
}

extern "C" int cssm_pf_init(cssm_pf* pf, double t0) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  int rc = cssm_launch_init(pf, t0);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

extern "C" int cssm_pf_init_from(cssm_pf* pf, double t0, const double* state_d) {
  if (!pf || !state_d) return fail(CSSM_EINVAL_ARG, "null argument");
  HIP_TRY(hipSetDevice(pf->device));
  HIP_TRY(hipMemcpyAsync(pf->d_m0, state_d, pf->d * 8, hipMemcpyHostToDevice, pf->stream));
  hipLaunchKernelGGL(k_init_from, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, pf->state[0], pf->stride, pf->n, pf->d, pf->d_m0);
  HIP_TRY(hipGetLastError());
  int rc = reset_scalars(pf);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(pf->stream));
  rc = upload_init_params(pf);   // d_m0 was used as scratch
  if (rc) return rc;
  fresh_host_state(pf, t0);
  return CSSM_OK;
}

extern "C" int cssm_pf_step(cssm_pf* pf, double t, double obs, int has_obs, double* ll_out, int32_t* ess_out) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (!pf->initialised) return fail(CSSM_ESTATE, "cssm_pf_step before cssm_pf_init");
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = cssm_ensure_recs(pf, 1);
  if (rc) return rc;
  cssm_build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[0]);
  rc = cssm_build_fsub(pf, 0, 1, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, 0, 1, true);
  if (rc) return rc;
  const int weighted = pf->h_recs[0].has_obs;
  const PreState before = pre_state(pf);
  pf->gen++;
  rc = launch_step(pf, pf->d_recs, weighted, pf->step);
  if (rc) return rc;
  Scalars& h = *pf->h_sc;
  rc = read_scalars(pf, 0, false, false, /*poll=*/true);   // (the completion word, not the stream: 40.4 -> 38.0 us per streaming step at N = 2^20)
  if (rc) return rc;
  if ((h.err & 64u) && !(h.err & 3u)) {   // the max ruled the reference level out: this observation again, relative to the max
    const uint32_t cleared = h.err & ~64u, none = 0xffffffffu;
    HIP_TRY(hipMemcpyAsync(&pf->sc->err, &cleared, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &none, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    rc = redo_observation(pf, before, pf->d_recs, nullptr, nullptr, 0);
    if (rc) return rc;
    rc = read_scalars(pf);
    if (rc) return rc;
  }
  pf->t = t; pf->step++;
  if (ll_out) *ll_out = h.ll;
  if (ess_out) *ess_out = h.ess;
  return cssm_check_device_err(pf, h);
}
// ---- an arbitrary Resample[A] on the host (model/package.scala:23): the step split at the resampler ----------------------
// cssm_pf_propagate = lines :117-124 of stepFilter (propagate, weigh); the caller fetches the proposed cloud and the
// log-weights (cssm_pf_get_proposed / cssm_pf_get_logw), applies ITS resampler to them and hands the result back with
// cssm_pf_adopt (:126-130).  A parity path -- the cloud crosses PCIe twice per observation -- for resamplers the library
// does not have natively (Resampling.indentity :29, user functions).
extern "C" int cssm_pf_propagate(cssm_pf* pf, double t, double obs, int has_obs) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (!pf->initialised) return fail(CSSM_ESTATE, "cssm_pf_propagate before cssm_pf_init");
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = cssm_ensure_recs(pf, 1);
  if (rc) return rc;
  cssm_build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[0]);
  rc = cssm_build_fsub(pf, 0, 1, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, 0, 1, false);   // (the level of this observation is the host's business)
  if (rc) return rc;
  pf->safe_sums = true;        // the host resampler wants the log-weights themselves (:123): the kernel that stores them
  rc = cssm_launch_propagate(pf, pf->d_recs);
  pf->safe_sums = false;
  if (rc) return rc;
  pf->wmode = false;
  // nobody decodes this step's running max on the device: clear every set of max slots AND of group sums for the next weighted
  // step (wparity restarts at 0 below: a native fused step that ran on set 0 before this call left its group sums there, and
  // k_offspring's publisher clears only the two sets it does not use -- the next native step would add onto them)
  static_assert(offsetof(Scalars, maxslot) == 0 && offsetof(Scalars, grp) == sizeof(Scalars::maxslot), "maxslot and grp are contiguous at the head of Scalars");
  HIP_TRY(hipMemsetAsync(pf->sc, 0, offsetof(Scalars, err), pf->stream));
  // ... and no kernel publishes this observation's max: the next native observation's level cannot be predicted from it (LGCP)
  static const double kNoLevel = cssm_nan();
  HIP_TRY(hipMemcpyAsync(&pf->sc->next_ref, &kNoLevel, sizeof(double), hipMemcpyHostToDevice, pf->stream));
  pf->have_level = false;
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  pf->wparity = 0;
  pf->t = t; pf->step++;
  return cssm_check_device_err(pf, h);
}

extern "C" int cssm_pf_adopt(cssm_pf* pf, const double* state_dN, double ll, int32_t ess) {
  if (!pf || !state_dN) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->initialised) return fail(CSSM_ESTATE, "cssm_pf_adopt before cssm_pf_init");
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  // the resampled cloud becomes the current one, in place of the proposed cloud (the next propagate reads it directly)
  HIP_TRY(hipMemcpy2DAsync(pf->state[pf->cur], pf->stride * 8, state_dN, pf->n * 8, pf->n * 8, pf->d, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(&pf->sc->ll, &ll, sizeof(double), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(&pf->sc->ess, &ess, sizeof(int32_t), hipMemcpyHostToDevice, pf->stream));
  const uint32_t no_pend = 0u;   // (an ESS still pending from an earlier native step is superseded by the caller's)
  HIP_TRY(hipMemcpyAsync(&pf->sc->pend, &no_pend, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  pf->ess_host = ess;
  HIP_TRY(hipStreamSynchronize(pf->stream));
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->src2 = nullptr; pf->anc_valid = false;
  return CSSM_OK;
}
// ------------------------------------------------------------------------------------ batch API

static int run_filter_once(cssm_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T, double* ll_out,
                           double* ll_t, int32_t* ess_t, double* path, bool* retry, bool cont = false);

// The series is enqueued without a host round trip per step, with the sums formed inside k_propagate relative to
// each observation's reference level.  If the max of some step rules its level out (err bit 2; an outlying
// observation), the series is run again with the sums in their own pass: every step then applies the same rule
// (cssm_ref_choose), so both runs define the same result and only the second one can compute it.
static int run_filter(cssm_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T, double* ll_out,
                      double* ll_t, int32_t* ess_t, double* path) {
  bool retry = false;
  int rc = run_filter_once(pf, t, y, has, T, ll_out, ll_t, ess_t, path, &retry);
  if (rc == CSSM_OK && retry) {
    pf->safe_sums = true;
    rc = run_filter_once(pf, t, y, has, T, ll_out, ll_t, ess_t, path, &retry);
    pf->safe_sums = false;
  }
  return rc;
}

// cont: the observations CONTINUE the filter from its current state (cssm_pf_ll_filter_more): no initial cloud is drawn, the
// first time increment is taken from the handle's clock, observation s of this call is observation pf->step + s of the filter.
static int run_filter_once(cssm_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T, double* ll_out,
                           double* ll_t, int32_t* ess_t, double* path, bool* retry, bool cont) {
  *retry = false;
  if (!pf || !t || !y) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data (the reference's minBy throws on an empty Vector)");
  // CSSM_CALL_TIMING=1: host-side phases of the call on stderr (tools/archive/leg_overhead.py)
  static const bool timing = getenv("CSSM_CALL_TIMING") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  double ph_rec = 0, ph_up = 0, ph_enq = 0;
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = cssm_ensure_recs(pf, T);
  if (rc) return rc;
  if (cont && (!pf->initialised || path)) return fail(CSSM_ESTATE, "continuing a filter needs an initialised handle (and records no path)");
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];     // data.minBy(_.t).t, model/ParticleFilter.scala:138
  const uint32_t base = cont ? pf->step : 0u;                  // index of this call's first observation in the filter's series
  double tp = cont ? pf->t : t0;
  // The first observation's record goes ahead of the others (a continued call of a model without per-observation tables): the queue
  // has its first kernels while the host is still building records 1 .. T-1 (4 us for 20 of them, on the critical path otherwise).
  const bool head_first = cont && T > 1 && pf->obs_kind != CSSM_OBS_LGCP;
  size_t built = 0;
  if (head_first) {
    cssm_build_rec(pf, tp, t[0], y[0], has ? has[0] : 1, base, &pf->h_recs[0]); tp = t[0]; built = 1;
    rc = cssm_upload_recs(pf, 0, 1, cont);
    if (rc) return rc;
  }
  bool rest_pending = head_first;
  auto build_rest = [&]() -> int {      // records built .. T-1: built and sent (head_first: behind the first observation's launches)
    for (size_t s = built; s < T; ++s) { cssm_build_rec(pf, tp, t[s], y[s], has ? has[s] : 1, base + (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
    int r = cssm_build_fsub(pf, 0, T, true);
    if (r) return r;
    ph_rec = since(tp0);
    r = cssm_upload_recs(pf, built, T - built, cont && built == 0);
    ph_up = since(tp0);
    rest_pending = false;
    return r;
  };
  if (!head_first) { rc = build_rest(); if (rc) return rc; }
  if (!cont) {
    rc = cssm_launch_init(pf, t0);
    if (rc) return rc;
  }
  // host-side state at the first observation of this call (launch_init: cur = 0, wparity = 0)
  const int cur0 = pf->cur, wpar0 = pf->wparity;
  const bool anc_valid0 = pf->anc_valid;
  const int32_t ess0 = pf->ess_host;      // what an unweighted first observation reports (ll and ess unchanged, :121)
  pf->gen++;
  const int d = pf->d;
  if (path) {
    if (pf->path_cap < (T + 1) * (size_t)d) {
      if (pf->d_path) (void)hipFree(pf->d_path);
      pf->d_path = nullptr;
      HIP_TRY(hipMalloc(&pf->d_path, (T + 1) * (size_t)d * 8));
      pf->path_cap = (T + 1) * (size_t)d;
    }
    const int32_t pr = (int32_t)cssm_philox_draw(pf->seed, 0, 0, CSSM_STREAM_PICK, 0).v[0];
    const uint32_t pa = pr < 0 ? (uint32_t)0 - (uint32_t)pr : (uint32_t)pr;
    hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride, (const uint32_t*)nullptr,
                       (uint64_t)pa % pf->n, d, pf->d_path);
  }
  if (pf->opt_events) HIP_TRY(hipEventRecord(pf->ev0, pf->stream));
  // path entry s + 1 = the resampled state sampleOne picks after observation s.  With the kernels that also form the
  // sums (small handles: the PMMH case) the k_propagate of observation s + 1, which gathers exactly that state into the
  // thread of slot pick_s, records it on the way; otherwise a one-block launch per observation does.
  // (whether an observation's propagate is such a kernel is decided observation by observation: an LGCP series starts with one
  //  whose level is its max -- `folded` = the propagate of the NEXT observation will record the pick after the current one)
  bool folded = false;
  // Per-observation kernels, enqueued without a host round trip.  With the sums formed inside k_propagate (relative to each
  // observation's reference level), an observation whose max rules its level out puts the series ON HOLD at that
  // observation (err bit 6: every kernel behind it returns at once); the host then redoes that one observation (its
  // propagate again, storing log-weights; sums relative to the max: k_tile_sums + k_offspring) and enqueues the rest again.
  size_t s_from = 0;
  for (;;) {
    const bool may_hold = may_use_sums_kernel(pf);
    for (size_t s = s_from; s < T; ++s) {
      const int weighted = pf->h_recs[s].has_obs;
      double* pick_out = (folded && s >= 1) ? pf->d_path + s * (size_t)d : nullptr;
      const uint32_t pick_slot = s >= 1 ? pf->h_recs[s - 1].pick : 0u;
      pf->h_step_for_resample = base + (uint32_t)s;
      rc = cssm_launch_propagate(pf, pf->d_recs + s, pick_out, pick_slot);
      if (!rc) {
        if (weighted) rc = launch_resample(pf, pf->d_recs + s, pf->d_ll_t, pf->d_ess_t, (uint32_t)s);
        else hipLaunchKernelGGL(k_record, dim3(1), dim3(1), 0, pf->stream, pf->sc, pf->d_ll_t, pf->d_ess_t, (uint32_t)s);
      }
      if (rc) return rc;
      if (rest_pending) { rc = build_rest(); if (rc) return rc; }
      folded = path && s + 1 < T && uses_sums_kernel(pf);   // (the last entry has no following propagate)
      if (path && !folded)
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride,
                           (const uint32_t*)(pf->anc_valid ? pf->anc : nullptr), (uint64_t)pf->h_recs[s].pick, d,
                           pf->d_path + (s + 1) * (size_t)d);
    }
    // the call's ONE synchronisation when no observation was held: results and scalars travel together
    ph_enq = since(tp0);
    if (pf->opt_events) HIP_TRY(hipEventRecord(pf->ev1, pf->stream));
    HIP_TRY(hipGetLastError());
    if (path) HIP_TRY(hipMemcpyAsync(path, pf->d_path, (T + 1) * (size_t)d * 8, hipMemcpyDeviceToHost, pf->stream));
    rc = read_scalars(pf, (uint32_t)T, ll_t != nullptr, ess_t != nullptr, /*poll=*/path == nullptr);   // (k_finish: a pending ESS formed, results into host-mapped memory)
    if (rc) return rc;
    const Scalars& hh = *pf->h_sc;
    if (!may_hold || !(hh.err & 64u) || (hh.err & 3u)) break;   // (NaN / unusable weights: reported below)
    const size_t sf = (size_t)hh.fail_step - base;              // (the record's index in this call)
    if (hh.fail_step < base || sf >= T) return fail(CSSM_ESTATE, "held series reports observation %u; this call holds %u .. %zu", hh.fail_step, base, base + T - 1);
    const uint32_t cleared = hh.err & ~64u, none = 0xffffffffu;
    HIP_TRY(hipMemcpyAsync(&pf->sc->err, &cleared, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &none, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    // host-side state right BEFORE the propagate of observation sf (every propagate flips cur, every weighted observation
    // advances the max-slot set; the ancestors are those of the observation before it, if that one resampled)
    PreState before;
    before.cur = (int)(((size_t)cur0 + sf) & 1);
    before.src = pf->state[before.cur]; before.src_stride = pf->stride;
    before.anc_valid = sf == 0 ? anc_valid0 : (pf->h_recs[sf - 1].has_obs != 0);
    int wp = wpar0;
    for (size_t q = 0; q < sf; ++q) wp = (wp + (pf->h_recs[q].has_obs ? 1 : 0)) % CSSM_MAXSETS;
    before.wparity = wp;
    pf->h_step_for_resample = base + (uint32_t)sf;
    rc = redo_observation(pf, before, pf->d_recs + sf, pf->d_ll_t, pf->d_ess_t, (uint32_t)sf);
    if (rc) return rc;
    folded = path && sf + 1 < T && uses_sums_kernel(pf);
    if (path && !folded)
      hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride, (const uint32_t*)pf->anc,
                         (uint64_t)pf->h_recs[sf].pick, d, pf->d_path + (sf + 1) * (size_t)d);
    s_from = sf + 1;             // (s_from == T: the loop only reads the results again)
  }
  Scalars& h = *pf->h_sc;
  if (ll_t) memcpy(ll_t, pf->h_ll_t, T * 8);
  if (ess_t) {
    memcpy(ess_t, pf->h_ess_t, T * 4);
    // an observation without a datum reports the ESS before it (:121), which a kernel could not know yet (the ESS of the
    // weighted observation before it was still pending when k_record ran)
    for (size_t s = 0; s < T; ++s) if (!pf->h_recs[s].has_obs) ess_t[s] = s ? ess_t[s - 1] : ess0;
  }
  pf->last_ms = -1.f;            // (the event pair is read when somebody asks: cssm_pf_last_loop_ms; a query per call cost ~3 us)
  pf->have_events = pf->opt_events != 0;
  if (timing && pf->have_events) { HIP_TRY(hipEventSynchronize(pf->ev1)); HIP_TRY(hipEventElapsedTime(&pf->last_ms, pf->ev0, pf->ev1)); }
  prof_collect(pf);
  if (timing) fprintf(stderr, "cssm call T=%zu: records built %.1f us, upload enqueued %.1f, %zu steps enqueued %.1f, results back %.1f (device loop %.1f us)\n",
                      T, ph_rec, ph_up, T, ph_enq, since(tp0), pf->last_ms * 1e3);
  pf->t = t[T - 1]; pf->step = base + (uint32_t)T;
  if ((h.err & 4u) && !(h.err & 1u) && !pf->safe_sums) { *retry = true; return CSSM_OK; }
  if (ll_out) *ll_out = h.ll;
  return cssm_check_device_err(pf, h);
}

extern "C" int cssm_pf_ll_filter(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                                 double* ll_out, double* ll_t, int32_t* ess_t) {
  return run_filter(pf, t, y, has_obs, T, ll_out, ll_t, ess_t, nullptr);
}

extern "C" int cssm_pf_ll_filter_more(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                                      double* ll_out, double* ll_t, int32_t* ess_t) {
  bool retry = false;
  int rc = run_filter_once(pf, t, y, has_obs, T, ll_out, ll_t, ess_t, nullptr, &retry, /*cont=*/true);
  if (rc == CSSM_OK && retry) return fail(CSSM_ESTATE, "a continued series cannot be repeated from its start");   // (not reached: batch series are held in place)
  return rc;
}

extern "C" int cssm_pf_filter(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                              double* ll_out, double* ll_t, int32_t* ess_t, double* path) {
  if (!path) return fail(CSSM_EINVAL_ARG, "path is null (use cssm_pf_ll_filter)");
  return run_filter(pf, t, y, has_obs, T, ll_out, ll_t, ess_t, path);
}
// ------------------------------------------------------------------------------------ contract diagnostics

__global__ void k_contract_eval(int fn, const double* __restrict__ x, size_t n, double* __restrict__ out, const double* __restrict__ logtab) {
  const double* tab = stage_log_table(logtab);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    if (fn == CSSM_FN_EXP) out[i] = cssm_exp(v);
    else if (fn == CSSM_FN_LOG) out[i] = cssm_log(v);
    else if (fn == CSSM_FN_LOG_UNIT) out[i] = cssm_log_unit(v, tab);
    else if (fn == CSSM_FN_SINCOS2PI) { double sn, cs; cssm_sincos2pi(v, &sn, &cs); out[2 * i] = sn; out[2 * i + 1] = cs; }
    else if (fn == CSSM_FN_SINCOS_U24) { double sn, cs; cssm_sincos_u24((uint32_t)v, tab, &sn, &cs); out[2 * i] = sn; out[2 * i + 1] = cs; }
    else if (fn == CSSM_FN_SQRT_RADIUS) out[i] = cssm_sqrt_radius(v);
    else { const cssm_u128 q = cssm_fix_from_double(v); out[2 * i] = cssm_u2d(q.lo); out[2 * i + 1] = cssm_u2d(q.hi); }
  }
}
template <int D>
__global__ void k_contract_normals(uint64_t seed, uint64_t first, uint32_t step, uint32_t tag, size_t n, double* __restrict__ out,
                                   const double* __restrict__ logtab) {
  const double* tab = stage_log_table(logtab);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double z[D];
    draw_normals<D>(seed, first + i, step, tag, tab, z);
#pragma unroll
    for (int k = 0; k < D; ++k) out[i * D + k] = z[k];
  }
}

extern "C" int cssm_contract_eval(int device, int fn, const double* x, size_t n, double* out, size_t n_out) {
  if (!x || !out || n < 1) return fail(CSSM_EINVAL_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(CSSM_EINVAL_ARG, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  size_t need = n, d = 0, np = 0;
  if (fn == CSSM_FN_SINCOS2PI || fn == CSSM_FN_FIX || fn == CSSM_FN_SINCOS_U24) need = 2 * n;
  else if (fn == CSSM_FN_PAIRED_NORMALS) {
    if (n < 5) return fail(CSSM_EINVAL_ARG, "paired normals take {seed, first, step, tag, d}");
    d = (size_t)x[4];
    if (d < 1 || d > CSSM_MAX_DIM) return fail(CSSM_EINVAL_ARG, "d out of range");
    np = n_out / d; need = np * d;
    if (np < 1) return fail(CSSM_EINVAL_ARG, "out is too small");
  } else if ((fn < CSSM_FN_EXP || fn > CSSM_FN_FIX) && fn != CSSM_FN_SINCOS_U24 && fn != CSSM_FN_SQRT_RADIUS) return fail(CSSM_EINVAL_ARG, "unknown contract function %d", fn);
  if (n_out < need) return fail(CSSM_EINVAL_ARG, "out holds %zu doubles, %zu needed", n_out, need);
  double *dx = nullptr, *dout = nullptr, *dtab = nullptr;
  int rc = CSSM_OK;
  if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dout, need * 8) != hipSuccess || hipMalloc(&dtab, sizeof(CSSM_TAB)) != hipSuccess)
    rc = fail(CSSM_ENOMEM, "hipMalloc");
  if (!rc && (hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dtab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice) != hipSuccess))
    rc = fail(CSSM_EHIP, "upload");
  if (!rc) {
    if (fn == CSSM_FN_PAIRED_NORMALS) {
      const uint64_t seed = (uint64_t)x[0], first = (uint64_t)x[1];
      const uint32_t step = (uint32_t)x[2], tag = (uint32_t)x[3];
      DISPATCH_D((int)d, k_contract_normals<D><<<dim3(grid_for(np, 256, 1024)), dim3(256)>>>(seed, first, step, tag, np, dout, dtab));
    } else {
      hipLaunchKernelGGL(k_contract_eval, dim3(grid_for(n, 256, 1024)), dim3(256), 0, 0, fn, dx, n, dout, dtab);
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = fail(CSSM_EHIP, "contract kernel");
  }
  if (!rc && hipMemcpy(out, dout, need * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CSSM_EHIP, "download");
  if (dx) (void)hipFree(dx);
  if (dout) (void)hipFree(dout);
  if (dtab) (void)hipFree(dtab);
  return rc;
}

// ---- on-box streaming ceiling (bench.py: roofline.copy_ceiling) ---------------------------------------------------
// A plain 16-bytes-per-lane copy of `bytes` bytes (read + write = 2 * bytes of HBM traffic), timed with HIP events over
// `reps` launches after one warm-up launch: what this GPU streams when nothing but loads and stores is in the way.
__global__ __launch_bounds__(256) void k_diag_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int cssm_diag_copy_ceiling(int device, size_t bytes, int reps, double* gbps_out) {
  if (!gbps_out || bytes < (1u << 20) || reps < 1) return fail(CSSM_EINVAL_ARG, "bytes >= 1 MiB, reps >= 1 and an output pointer are required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(CSSM_EINVAL_ARG, "device %d out of range", device);
  HIP_TRY(hipSetDevice(device));
  const size_t n16 = bytes / 16;
  uint4 *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = CSSM_OK;
  if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&b, n16 * 16) != hipSuccess) rc = fail(CSSM_ENOMEM, "hipMalloc of 2 x %zu bytes", n16 * 16);
  if (!rc && (hipMemset(a, 1, n16 * 16) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail(CSSM_EHIP, "setup");
  if (!rc) {
    const int grid = 65536;      // many short blocks: measured best for a plain copy (tools/copy_bench.hip: 5.4 TB/s at 1 GiB, 4.7 with 4096 blocks)
    hipLaunchKernelGGL(k_diag_copy, dim3(grid), dim3(256), 0, 0, a, b, n16);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_diag_copy, dim3(grid), dim3(256), 0, 0, (r & 1) ? b : a, (r & 1) ? a : b, n16);
    (void)hipEventRecord(e1, 0);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(CSSM_EHIP, "copy kernel");
    else *gbps_out = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  return rc;
}

extern "C" int cssm_pf_last_loop_ms(cssm_pf* pf, float* ms_out) {
  if (!pf || !ms_out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (pf->last_ms < 0.f) {       // the batch drivers leave the event pair of their device loop unread
    if (!pf->have_events) return fail(CSSM_ESTATE, "the last batch call recorded no events: set CSSM_OPT_LOOP_EVENTS = 1 before it");
    HIP_TRY(hipSetDevice(pf->device));
    HIP_TRY(hipEventSynchronize(pf->ev1));   // (the call returned on k_finish's completion word, not on the stream)
    HIP_TRY(hipEventElapsedTime(&pf->last_ms, pf->ev0, pf->ev1));
  }
  *ms_out = pf->last_ms;
  return CSSM_OK;
}

extern "C" int cssm_pf_last_device_us(cssm_pf* pf, double* us_out) {
  if (!pf || !us_out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->h_sc) return fail(CSSM_ESTATE, "no call has run");
  const unsigned long long a = pf->h_sc->t_first, b = pf->h_sc->t_last;
  if (a == 0ull || b < a) return fail(CSSM_ESTATE, "the last call left no pair of device stamps (it drew a new cloud, or none has run)");
  *us_out = (double)(b - a) * 1e-2;   // 100 MHz
  return CSSM_OK;
}

extern "C" int cssm_pf_stream_idle(cssm_pf* pf) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (hipSetDevice(pf->device) != hipSuccess) return fail(CSSM_EHIP, "hipSetDevice(%d)", pf->device);
  const hipError_t q = hipStreamQuery(pf->stream);
  if (q == hipSuccess) return 1;
  if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return fail(CSSM_EHIP, "hipStreamQuery: %s", hipGetErrorString(q));
}

extern "C" int cssm_pf_set_option(cssm_pf* pf, int option, int value) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (option == CSSM_OPT_EXACT_OFFSPRING) { pf->opt_exact = (value == 2) ? 2 : (value ? 1 : 0); return CSSM_OK; }
  if (option == CSSM_OPT_FUSED_SUMS) { pf->opt_fused = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_GROUP_SUMS) { pf->opt_grp = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_WAVE_SUMS) { pf->opt_wave = value == 2 ? 2 : (value ? 1 : 0); return CSSM_OK; }
  if (option == CSSM_OPT_LOOP_EVENTS) { pf->opt_events = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_SPECIALISE) { pf->opt_spec = (value == 2) ? 2 : (value ? 1 : 0); return CSSM_OK; }
  if (option == CSSM_OPT_WHOLE_TILES) {   // launch geometry only: the arrays hold up to four sub-units per unit either way
    if (pf->sharded) return fail(CSSM_ESTATE, "sharded handles always run whole tiles");
    pf->opt_whole = value < 0 ? 0 : (value > 3 ? 3 : value);
    pf->split = value ? 1u : auto_split(pf);
    return CSSM_OK;
  }
  if (option == CSSM_OPT_RESAMPLER) {
    if (value < CSSM_RESAMPLE_SYSTEMATIC || value > CSSM_RESAMPLE_MULTINOMIAL) return fail(CSSM_EINVAL_ARG, "unknown resampler %d", value);
    // (a sharded multinomial resampler is another exchange altogether: slot i draws its own uniform, so a rank's slots take their
    //  ancestors from every rank -- a general gather, not the boundary rows of two neighbours)
    if (pf->sharded && value == CSSM_RESAMPLE_MULTINOMIAL) return fail(CSSM_ESTATE, "sharded handles resample systematically or stratified");
    pf->resampler = value;
    return CSSM_OK;
  }
  return fail(CSSM_EINVAL_ARG, "unknown option %d", option);
}

extern "C" int cssm_pf_profile(cssm_pf* pf, int enable) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  pf->profile = enable != 0;
  pf->prof_used = 0;
  for (int k = 0; k < CSSM_NKERNELS; ++k) { pf->prof_ms[k] = 0.0; pf->prof_cnt[k] = 0; }
  return CSSM_OK;
}

extern "C" int cssm_pf_profile_read(cssm_pf* pf, double* total_ms, uint64_t* launches) {
  if (!pf || !total_ms || !launches) return fail(CSSM_EINVAL_ARG, "null argument");
  for (int k = 0; k < CSSM_NKERNELS; ++k) { total_ms[k] = pf->prof_ms[k]; launches[k] = pf->prof_cnt[k]; }
  return CSSM_OK;
}

// ------------------------------------------------------------------------------------ inspection

extern "C" int cssm_pf_get_particles(cssm_pf* pf, double* out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->initialised) return fail(CSSM_ESTATE, "not initialised");
  HIP_TRY(hipSetDevice(pf->device));
  double* tmp = nullptr;
  if (hipMalloc(&tmp, pf->n * (size_t)pf->d * 8) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc for the gathered cloud");
  hipLaunchKernelGGL(k_gather, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, pf->src, pf->src_stride,
                     (const uint32_t*)(pf->anc_valid ? pf->anc : nullptr), tmp, (size_t)pf->n, pf->n, pf->d,
                     pf->anc_valid ? pf->src2 : nullptr, pf->src2_stride, pf->n_split);
  hipError_t e = hipMemcpyAsync(out, tmp, pf->n * (size_t)pf->d * 8, hipMemcpyDeviceToHost, pf->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(pf->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(CSSM_EHIP, "get_particles: %s", hipGetErrorString(e));
  return CSSM_OK;
}

extern "C" int cssm_pf_get_proposed(cssm_pf* pf, double* out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->initialised) return fail(CSSM_ESTATE, "not initialised");
  HIP_TRY(hipSetDevice(pf->device));
  HIP_TRY(hipMemcpy2DAsync(out, pf->n * 8, pf->state[pf->cur], pf->stride * 8, pf->n * 8, pf->d, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

extern "C" int cssm_pf_get_ancestors(cssm_pf* pf, uint32_t* out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
  HIP_TRY(hipSetDevice(pf->device));
  if (!pf->anc_valid) { for (uint64_t i = 0; i < pf->n; ++i) out[i] = (uint32_t)i; return CSSM_OK; }
  HIP_TRY(hipMemcpyAsync(out, pf->anc, pf->n * 4, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

extern "C" int cssm_pf_get_weights(cssm_pf* pf, double* out, double* level_out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->wmode) return fail(CSSM_ESTATE, "the last weighted step kept log-weights (cssm_pf_get_logw), not weights");
  HIP_TRY(hipSetDevice(pf->device));
  Scalars h;
  HIP_TRY(hipMemcpyAsync(out, pf->logw, pf->n * 8, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (level_out) *level_out = h.ref;
  return CSSM_OK;
}

extern "C" int cssm_pf_get_logw(cssm_pf* pf, double* out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (pf->wmode) return fail(CSSM_ESTATE, "the fused kernel keeps the weights exp(w - c) in place of the log-weights (cssm_pf_get_weights); "
                                          "CSSM_OPT_FUSED_SUMS = 0 keeps log-weights");
  HIP_TRY(hipSetDevice(pf->device));
  HIP_TRY(hipMemcpyAsync(out, pf->logw, pf->n * 8, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

// getIntervals (model/ParticleFilter.scala:415-424) on the device; see include/cssm_pf.h
static double host_link(int obs_kind, double g) {
  switch (obs_kind) {
    case CSSM_OBS_POISSON: case CSSM_OBS_NEGBIN: case CSSM_OBS_ZIP: return cssm_exp(g);
    case CSSM_OBS_BERNOULLI: return (g > 6.0) ? 1.0 : ((g < -6.0) ? 0.0 : 1.0 / (1.0 + cssm_exp(-g)));
    case CSSM_OBS_BETA: return cssm_exp(-g);
    default: return g;
  }
}

double cssm_eta_of_mean(const cssm_pf* pf, const StepRec& rec, const double* mean) {
  double g = 0.0, acc = 0.0;
  for (int k = 0; k < pf->d; ++k) {
    const int fm = pf->mk.fmode(k);
    if (fm == FM_START) acc = rec.fco[k] * mean[k]; else if (fm == FM_ADD) acc = acc + rec.fco[k] * mean[k];
    if (pf->mk.leaf_end(k)) g = pf->mk.first_leaf(k) ? acc : g + acc;
  }
  return host_link(pf->obs_kind, g);
}

// Summary of the cloud { src[:, idx[i]] : i < n } at time `time` (idx == nullptr: identity); see cssm_pf_summary.
static int summary_impl(cssm_pf* pf, const double* src, size_t src_stride, const uint32_t* idx, const double* src2, size_t src2_stride,
                        uint32_t n_split, double time, double interval, double* state_mean, double* state_lower, double* state_upper,
                        double* eta_of_mean, double* eta_lower, double* eta_upper) {
  const int d = pf->d, rows = d + 1;
  const uint64_t n = pf->n;
  const int nblocks = grid_for(n, CSSM_BLOCK, 1024);
  unsigned long long* keys = nullptr; double* partial = nullptr; SelState* st = nullptr; uint32_t* hist = nullptr; double* out = nullptr;
  StepRec* drec = nullptr;
  std::vector<SelState> hst(rows);
  std::vector<double> hout(3 * rows);
  StepRec hrec;
  int rc = CSSM_OK;
  // ranks, 0-based in ascending order: getCredibleInterval (:488-502) uses (N - index - 1, index - 1) with
  // index = floor(interval * N); getOrderStatistic (:455-460) uses (N - index, index) -- both reproduced
  const long long idxr = (long long)std::floor(interval * (double)n);
  auto clampr = [&](long long r) { return (unsigned long long)std::min<long long>(std::max<long long>(r, 0), (long long)n - 1); };
  for (int k = 0; k < rows; ++k) {
    hst[k].prefix[0] = hst[k].prefix[1] = 0;
    hst[k].rank[0] = clampr(k < d ? (long long)n - idxr - 1 : (long long)n - idxr);
    hst[k].rank[1] = clampr(k < d ? idxr - 1 : idxr);
  }
  cssm_build_rec(pf, time, time, 0.0, 0, pf->step, &hrec);   // F(t) of the requested time for f(x, t)
#define SM_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(CSSM_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); goto done; } } while (0)
  SM_TRY(hipMalloc(&keys, (size_t)rows * n * 8)); SM_TRY(hipMalloc(&partial, (size_t)nblocks * d * 8));
  SM_TRY(hipMalloc(&st, rows * sizeof(SelState))); SM_TRY(hipMalloc(&hist, (size_t)rows * 512 * 4)); SM_TRY(hipMalloc(&out, 3 * rows * 8));
  SM_TRY(hipMalloc(&drec, sizeof(StepRec)));
  SM_TRY(hipMemcpyAsync(st, hst.data(), rows * sizeof(SelState), hipMemcpyHostToDevice, pf->stream));
  SM_TRY(hipMemcpyAsync(drec, &hrec, sizeof hrec, hipMemcpyHostToDevice, pf->stream));
  SM_TRY(hipMemsetAsync(hist, 0, (size_t)rows * 512 * 4, pf->stream));
  {
    DISPATCH_D(d, k_summary_fill<D><<<dim3(nblocks), dim3(CSSM_BLOCK), 0, pf->stream>>>(
                      src, src_stride, idx, idx ? src2 : nullptr, src2_stride, n_split, n, drec, pf->mk, keys, (size_t)n, partial));
    for (int shift = 56; shift >= 0; shift -= 8) {
      hipLaunchKernelGGL(k_sel_hist, dim3(nblocks, rows), dim3(CSSM_BLOCK), 0, pf->stream, keys, (size_t)n, n, st, shift, hist);
      hipLaunchKernelGGL(k_sel_pick, dim3(rows), dim3(2), 0, pf->stream, st, shift, hist);
    }
    hipLaunchKernelGGL(k_summary_finish, dim3(1), dim3(64), 0, pf->stream, st, partial, nblocks, d, n, out);
  }
  SM_TRY(hipGetLastError());
  SM_TRY(hipMemcpyAsync(hout.data(), out, 3 * rows * 8, hipMemcpyDeviceToHost, pf->stream));
  SM_TRY(hipStreamSynchronize(pf->stream));
  for (int k = 0; k < d; ++k) {
    if (state_mean) state_mean[k] = hout[k];
    if (state_lower) state_lower[k] = hout[rows + k];
    if (state_upper) state_upper[k] = hout[2 * rows + k];
  }
  if (eta_lower) *eta_lower = hout[rows + d];
  if (eta_upper) *eta_upper = hout[2 * rows + d];
  if (eta_of_mean) *eta_of_mean = cssm_eta_of_mean(pf, hrec, hout.data());   // meanEta = link(f(stateMean, t)), :420
done:
#undef SM_TRY
  { void* ptrs[] = {keys, partial, st, hist, out, drec}; for (void* q : ptrs) if (q) (void)hipFree(q); }
  return rc;
}

extern "C" int cssm_pf_summary(cssm_pf* pf, double interval, double* state_mean, double* state_lower, double* state_upper,
                               double* eta_of_mean, double* eta_lower, double* eta_upper) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (!pf->initialised) return fail(CSSM_ESTATE, "not initialised");
  if (!(interval > 0.0 && interval <= 1.0)) return fail(CSSM_EINVAL_ARG, "interval must be in (0, 1]");
  HIP_TRY(hipSetDevice(pf->device));
  return summary_impl(pf, pf->src, pf->src_stride, pf->anc_valid ? pf->anc : nullptr, pf->src2, pf->src2_stride, pf->n_split, pf->t,
                      interval, state_mean, state_lower, state_upper, eta_of_mean, eta_lower, eta_upper);
}

// ------------------------------------------------------------------------------------ FilterInterpolate
// FilterInterpolate.stepInterpolate / filterInterpolate (model/ParticleFilter.scala:273-311): particles are whole
// paths (x_t :: x_{t-1} :: ...), and a weighted step resamples the PATHS.  Here the forward pass keeps every
// propagated cloud and ancestor array in a history slab (the kernels write straight into it), and the paths that
// survive to the end are recovered by composing the ancestor arrays backwards: the state at time index s of final
// path i is X1_s[b_s(i)], b_T = anc_T, b_{s-1} = anc_{s-1}[b_s].  Output: per time index the summary
// examples/Interpolate.scala:42-44 computes from the transposed paths (getIntervals of the surviving lineages).

__global__ void k_compose(const uint32_t* __restrict__ anc, uint32_t* __restrict__ b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = anc[b[i]];
}
__global__ void k_iota(uint32_t* __restrict__ b, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = (uint32_t)i;
}

extern "C" int cssm_pf_interpolate(cssm_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T, double interval,
                                   int flags, double* ll_out, double* state_mean, double* state_lower, double* state_upper,
                                   double* eta_of_mean, double* eta_lower, double* eta_upper) {
  if (!pf || !t || !y) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data");
  if (pf->sharded) return fail(CSSM_ESTATE, "interpolation runs on a single-GPU handle");
  if (pf->obs_kind == CSSM_OBS_LGCP) return fail(CSSM_EINVAL_ARG, "FilterInterpolate weighs with dataLikelihood; the LGCP filter has no path variant");
  if (!(interval > 0.0 && interval <= 1.0)) return fail(CSSM_EINVAL_ARG, "interval must be in (0, 1]");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = cssm_ensure_recs(pf, T);
  if (rc) return rc;
  const int d = pf->d;
  const size_t slab = pf->stride * (size_t)d;
  double* hx = nullptr; uint32_t *hanc = nullptr, *bidx = nullptr;
  std::vector<uint8_t> weighted(T + 1, 0);
  double* save_state[2] = {pf->state[0], pf->state[1]};
  uint32_t* save_anc = pf->anc;
  if (hipMalloc(&hx, (T + 1) * slab * 8 + 64) != hipSuccess) return fail(CSSM_ENOMEM, "history of %zu clouds does not fit (%zu bytes)", T + 1, (T + 1) * slab * 8);
  if (hipMalloc(&hanc, (T + 1) * pf->stride * 4) != hipSuccess || hipMalloc(&bidx, pf->stride * 4) != hipSuccess) {
    (void)hipFree(hx); if (hanc) (void)hipFree(hanc);
    return fail(CSSM_ENOMEM, "ancestor history does not fit");
  }
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];
  double tp = t0;
  for (size_t s = 0; s < T; ++s) { cssm_build_rec(pf, tp, t[s], y[s], has ? has[s] : 1, (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = cssm_build_fsub(pf, 0, T, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, 0, T, false);
  Scalars h;
  for (int attempt = 0; attempt < 2 && !rc; ++attempt) {   // second attempt: see run_filter
    pf->safe_sums = (attempt == 1);
    pf->state[0] = hx;                                   // X1_0 = the initial cloud
    rc = cssm_launch_init(pf, t0);
    for (size_t s = 0; s < T && !rc; ++s) {
      // step s+1 reads X1_s through anc_s (launch_propagate uses pf->anc / pf->anc_valid) and writes X1_{s+1}
      pf->state[pf->cur ^ 1] = hx + (s + 1) * slab;
      pf->anc = hanc + s * pf->stride;
      const int w = pf->h_recs[s].has_obs;
      pf->h_step_for_resample = (uint32_t)s;
      rc = cssm_launch_propagate(pf, pf->d_recs + s);
      if (!rc && w) {
        pf->anc = hanc + (s + 1) * pf->stride;
        rc = launch_resample(pf, pf->d_recs + s);
        weighted[s + 1] = 1;
      }
    }
    if (!rc && hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream) != hipSuccess) rc = fail(CSSM_EHIP, "scalars");
    if (!rc && hipStreamSynchronize(pf->stream) != hipSuccess) rc = fail(CSSM_EHIP, "forward pass");
    if (rc || !(h.err & 64u) || (h.err & 3u)) break;   // (bit 6: some observation's level was ruled out -- again, sums in their own pass)
  }
  pf->safe_sums = false;
  if (!rc) rc = cssm_check_device_err(pf, h);
  if (!rc && ll_out) *ll_out = h.ll;
  // backward: compose the genealogy and summarise every time index
  if (!rc) {
    hipLaunchKernelGGL(k_iota, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, bidx, pf->n);
    for (size_t s = T + 1; s-- > 0 && !rc;) {
      if (weighted[s])
        hipLaunchKernelGGL(k_compose, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, hanc + s * pf->stride, bidx, pf->n);
      // output entry o pairs the cloud of time index s with the time of index o (o = T - s reproduces the zipped
      // pairing of examples/Interpolate.scala:42, whose transposed paths run newest-first)
      const size_t o = (flags & CSSM_INTERP_REFERENCE_PAIRING) ? T - s : s;
      const double time = (o == 0) ? t0 : t[o - 1];
      rc = summary_impl(pf, hx + s * slab, pf->stride, bidx, nullptr, 0, 0, time, interval,
                        state_mean ? state_mean + o * d : nullptr, state_lower ? state_lower + o * d : nullptr,
                        state_upper ? state_upper + o * d : nullptr, eta_of_mean ? eta_of_mean + o : nullptr,
                        eta_lower ? eta_lower + o : nullptr, eta_upper ? eta_upper + o : nullptr);
    }
  }
  (void)hipStreamSynchronize(pf->stream);
  pf->state[0] = save_state[0]; pf->state[1] = save_state[1]; pf->anc = save_anc;
  pf->initialised = false;                               // the handle's own buffers hold no cloud now
  (void)hipFree(hx); (void)hipFree(hanc); (void)hipFree(bidx);
  pf->t = t[T - 1]; pf->step = (uint32_t)T;
  return rc;
}

// ------------------------------------------------------------------------------------ stateless resampler

extern "C" int cssm_resample_systematic(const double* w, size_t n, double u, uint32_t* anc, int device) {
  return cssm_resample(CSSM_RESAMPLE_SYSTEMATIC, w, n, u, 0, 0, anc, device);
}

// kind = CSSM_RESAMPLE_*: systematic reads u; stratified and multinomial draw their per-slot uniforms from the Philox
// streams of (seed, step) exactly as the filter's resamplers do at observation `step`
extern "C" int cssm_resample(int kind, const double* w, size_t n, double u, uint64_t seed, uint32_t step, uint32_t* anc, int device) {
  if (!w || !anc) return fail(CSSM_EINVAL_ARG, "null argument");
  if (n < 1 || n >= 0xffffffffull) return fail(CSSM_EINVAL_ARG, "n out of range");
  if (kind < CSSM_RESAMPLE_SYSTEMATIC || kind > CSSM_RESAMPLE_MULTINOMIAL) return fail(CSSM_EINVAL_ARG, "unknown resampler %d", kind);
  if (kind != CSSM_RESAMPLE_SYSTEMATIC) u = 0.0;
  if (!(u >= 0.0 && u < 1.0)) return fail(CSSM_EINVAL_ARG, "u must be in [0, 1)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  HIP_TRY(hipSetDevice(device));
  const uint32_t ntiles = (uint32_t)((n + CSSM_TILE - 1) / CSSM_TILE);
  const uint32_t sup = (ntiles + 1023u) / 1024u, nunits = (ntiles + sup - 1) / sup;
  const size_t stride = (size_t)ntiles * CSSM_TILE;
  double* d_w = nullptr; uint32_t *d_end = nullptr, *d_anc = nullptr; cssm_u128 *tS = nullptr, *tS2 = nullptr, *tP = nullptr;
  Scalars* sc = nullptr; StepRec* d_rec = nullptr; double* d_tab = nullptr; double* d_cum = nullptr;
  hipStream_t st = nullptr;
  int rc = CSSM_OK;
  StepRec hrec; memset(&hrec, 0, sizeof hrec); hrec.u = u; hrec.step = step;
  Scalars hs;
  std::vector<double> wscaled;
#define RS_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(CSSM_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); goto done; } } while (0)
  RS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  RS_TRY(hipMalloc(&d_w, stride * 8)); RS_TRY(hipMalloc(&d_end, stride * 4)); RS_TRY(hipMalloc(&d_anc, stride * 4));
  RS_TRY(hipMalloc(&tS, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tS2, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tP, ntiles * sizeof(cssm_u128)));
  RS_TRY(hipMalloc(&sc, sizeof(Scalars))); RS_TRY(hipMalloc(&d_rec, sizeof(StepRec))); RS_TRY(hipMalloc(&d_tab, sizeof(CSSM_TAB)));
  if (kind == CSSM_RESAMPLE_MULTINOMIAL) RS_TRY(hipMalloc(&d_cum, stride * 8));
  RS_TRY(hipMemcpyAsync(d_tab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice, st));
  RS_TRY(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
  {
    // Weights of ANY scale: the reference normalises w / sum(w) (Resampling.scala:21-24) and its own property is stated over every
    // non-empty vector in [0, 1] (SamplingTest.scala:12-22), but the contract's sums live on a 2^-96 grid -- a vector whose largest
    // weight is below 2^-32 (inside a filter the largest is ~1) is brought up by the exact power of two that puts that weight into
    // [0.5, 1) before it is summed.  The oracle's seam does the same (oracle/oracle.py: seam_scale).
    double wmax = 0.0;
    for (size_t i = 0; i < n; ++i) if (w[i] > wmax) wmax = w[i];
    if (wmax > 0.0 && wmax < 0x1.0p-32) {
      int e = 0;
      (void)std::frexp(wmax, &e);
      wscaled.resize(n);
      for (size_t i = 0; i < n; ++i) wscaled[i] = std::ldexp(w[i], -e);
      w = wscaled.data();
    }
  }
  RS_TRY(hipMemcpyAsync(d_w, w, n * 8, hipMemcpyHostToDevice, st));
  RS_TRY(hipMemcpyAsync(d_rec, &hrec, sizeof hrec, hipMemcpyHostToDevice, st));
  {
    const int tgrid = (int)nunits;
    hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, st, d_w, (uint64_t)n, sc, tS, tS2, ntiles, sup, nunits, 1, -1, (const double*)nullptr, d_tab,
                       (const StepRec*)d_rec, 0u);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tS, tS2, tP, nunits, sc, (uint64_t)n, 1, (double*)nullptr, (int32_t*)nullptr, 0u,
                       (const double*)nullptr, (unsigned long long*)nullptr, 0, 0u, 1);
#define RS_OFF_ARGS d_w, (uint64_t)n, sc, (const cssm_u128*)tP, (const cssm_u128*)tS2, d_rec, (uint64_t)n, d_end, d_anc, ntiles, sup, nunits, 1, 0, \
                    (double*)nullptr, (int32_t*)nullptr, 0u, 0, (const unsigned long long*)nullptr, 0, 1, 1, seed, d_cum, d_tab, 0,                \
                    (unsigned long long*)nullptr, 0u, (uint32_t)n
    if (kind == CSSM_RESAMPLE_STRATIFIED)
      hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_STRATIFIED>), dim3(tgrid), dim3(CSSM_BLOCK), 0, st, RS_OFF_ARGS);
    else if (kind == CSSM_RESAMPLE_MULTINOMIAL)
      hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_MULTINOMIAL>), dim3(tgrid), dim3(CSSM_BLOCK), 0, st, RS_OFF_ARGS);
    else
      hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_SYSTEMATIC>), dim3(tgrid), dim3(CSSM_BLOCK), 0, st, RS_OFF_ARGS);
#undef RS_OFF_ARGS
    if (kind == CSSM_RESAMPLE_MULTINOMIAL)
      hipLaunchKernelGGL(k_multinomial, dim3(grid_for(n, 256, kGridCap)), dim3(256), 0, st, d_cum, (uint64_t)n, seed, step, d_anc);
  }
  RS_TRY(hipGetLastError());
  RS_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&hs, sc), hipMemcpyDeviceToHost, st));
  RS_TRY(hipMemcpyAsync(anc, d_anc, n * 4, hipMemcpyDeviceToHost, st));
  RS_TRY(hipStreamSynchronize(st));
  if (hs.S_tot.lo == 0 && hs.S_tot.hi == 0) rc = fail(CSSM_ENONFINITE, "all weights are zero (the reference divides by a zero total)");
done:
#undef RS_TRY
  void* ptrs[] = {d_w, d_end, d_anc, tS, tS2, tP, sc, d_rec, d_tab, d_cum};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

#ifdef CSSM_OFF_STAMPS
// diagnostic build: the stamps k_offspring_self's blocks left (8 words per block)
extern "C" int cssm_pf_debug_stamps(cssm_pf* pf, unsigned long long* out, size_t nwords) {
  if (!pf || !pf->cum || nwords * 8 > (1u << 20)) return CSSM_EINVAL_ARG;
  HIP_TRY(hipStreamSynchronize(pf->stream));
  HIP_TRY(hipMemcpy(out, pf->cum, nwords * 8, hipMemcpyDeviceToHost));
  return CSSM_OK;
}
#endif

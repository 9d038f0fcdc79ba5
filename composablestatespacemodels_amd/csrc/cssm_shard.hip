// cssm_shard.hip -- the sharded filter of libcssm_pf: the cssm_pf_shard_* stage calls between which the caller runs its
// collectives, and the series loop in which the library issues them itself (RCCL over xGMI, resolved at run time).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "cssm_internal.h"
#include "cssm_shard_kernels.hip.h"

// ------------------------------------------------------------------------------------ sharded stages
// One process per GPU; the collectives between the stages belong to the caller (RCCL through
// torch.distributed).  See include/cssm_pf.h for the sequence.

// level of the step from the all-gathered order keys (word 4 of every rank's 5 words)
__global__ void k_import_level(Scalars* sc, const unsigned long long* __restrict__ all5, int world, const StepRec* __restrict__ rec) {
  if (sc->err & (4u | 8u | 16u)) return;   // on hold / void: the level of the observation the series holds at stays in place
  unsigned long long key = 0ull;
  for (int r = 0; r < world; ++r) { const unsigned long long k = all5[5 * r + 4]; key = (k > key) ? k : key; }
  sc->gmax = cssm_order_unkey(key);
  sc->ref = cssm_ref_choose(rec->ref, sc->gmax);   // the level every kernel of the step agrees on (k_tile_sums applies the same rule)
}

// For every destination rank q (owner of slots [q*n_per, min((q+1)*n_per, N))): the contiguous
// range of LOCAL particles that own at least one of q's slots.  One thread per q.
__global__ void k_send_ranges(const uint32_t* __restrict__ endslot, uint64_t n_local, const Scalars* __restrict__ sc,
                              const StepRec* __restrict__ rec, uint64_t n_global, int rank, int world, uint64_t n_per,
                              long long* __restrict__ first, long long* __restrict__ count, int rs, uint64_t seed) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= world) return;
  uint64_t b_lo = (uint64_t)q * n_per, b_hi = b_lo + n_per;
  if (b_lo > n_global) b_lo = n_global;
  if (b_hi > n_global) b_hi = n_global;
  uint64_t e_before = 0;   // end slot of the last particle of the previous rank
  if (rank > 0) {
    const double Cb = cssm_u128_to_double(sc->S_off) / cssm_u128_to_double(sc->S_tot);
    e_before = (rs == CSSM_RESAMPLE_STRATIFIED) ? cssm_strat_count(Cb, seed, rec->step, n_global) : cssm_sys_count(Cb, rec->u, n_global);
  }
  // j_lo = first local j with endslot[j] > b_lo
  uint64_t lo = 0, hi = n_local;
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)endslot[mid] > b_lo) hi = mid; else lo = mid + 1; }
  const uint64_t j_lo = lo;
  const uint64_t start = (j_lo == 0) ? e_before : (uint64_t)endslot[j_lo - 1];
  if (b_lo >= b_hi || j_lo >= n_local || start >= b_hi) { first[q] = 0; count[q] = 0; return; }
  // j_last = first local j with endslot[j] >= b_hi (it owns slot b_hi - 1), clamped
  lo = j_lo; hi = n_local;
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)endslot[mid] >= b_hi) hi = mid; else lo = mid + 1; }
  uint64_t j_last = lo;
  if (j_last >= n_local) j_last = n_local - 1;
  first[q] = (long long)j_lo;
  count[q] = (long long)(j_last - j_lo + 1);
}

// rows of d+1 doubles (the particle's state and its end slot) for every destination rank, destinations
// back to back: row r belongs to the destination q with sum(count[<q]) <= r < sum(count[<=q])
__global__ void k_pack(const double* __restrict__ src, size_t stride, const uint32_t* __restrict__ endslot, int d, int world,
                       const long long* __restrict__ first, const long long* __restrict__ count, long long total, int skip,
                       double* __restrict__ out) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long long)gridDim.x * blockDim.x) {
    long long off = 0;
    int q = 0;
    for (;;) {   // destination of row r; the rank's own range (`skip`) never travels
      const long long c = (q == skip) ? 0 : count[q];
      if (r < off + c || q == world - 1) break;
      off += c; ++q;
    }
    const long long j = first[q] + (r - off);
    double* row = out + r * (d + 1);
    for (int k = 0; k < d; ++k) row[k] = src[(size_t)k * stride + (size_t)j];
    row[d] = (double)endslot[j];
  }
}
// Received rows (d+1 doubles: state, end slot) of the ranks below (first n_low rows) and above this one, in global
// particle order: states go to the SoA candidate buffer, end slots and state indices to the candidate lists.
__global__ void k_adopt_remote(const double* __restrict__ recv, long long m, int d,
                               uint32_t n_split, double* __restrict__ cand, size_t cstride,
                               uint32_t* __restrict__ cand_end, uint32_t* __restrict__ cand_idx) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x) {
    const double* row = recv + r * (d + 1);
    for (int k = 0; k < d; ++k) cand[(size_t)k * cstride + (size_t)r] = row[k];
    cand_end[r] = (uint32_t)row[d];
    cand_idx[r] = n_split + (uint32_t)r;
  }
}

static int shard_prepare_step(cssm_pf* pf, const StepRec* d_rec, int weighted, uint64_t* sums5_dev);
static int bounded_sync(cssm_pf* pf);
static int shard_check(cssm_pf* pf);

// ---- `filter` on shards (model/ParticleFilter.scala:152-158): after every observation sampleOne picks ONE particle of the
// whole cloud, uniformly by GLOBAL slot (Resampling.scala:151-154; the index is a function of seed and observation: every rank
// computes the same one) -- the rank that owns the slot records the state it holds, rows of the others stay zero, and the
// caller combines the ranks' paths (exactly one of them is non-zero per row).
__global__ void k_pick_shard(const double* __restrict__ src, size_t src_stride, const uint32_t* __restrict__ anc, const double* __restrict__ src2,
                             size_t src2_stride, uint32_t n_split, uint64_t idx, int d, double* __restrict__ out_row) {
  const int k = threadIdx.x;
  if (k >= d) return;
  const size_t j = anc ? (size_t)anc[idx] : (size_t)idx;
  out_row[k] = (src2 && j >= n_split)
      ? (src2_stride == 0 ? ld_sys_f64(src2 + (size_t)(j - n_split) * (size_t)(d + 1) + k) : ld_sys_f64(src2 + (size_t)k * src2_stride + (j - n_split)))
      : src[(size_t)k * src_stride + j];
}
static int path_prepare(cssm_pf* pf, size_t T) {
  if (!pf->want_path) return CSSM_OK;
  const size_t need = (T + 1) * (size_t)pf->d;
  if (pf->path_cap < need) {
    HIP_TRY(hipStreamSynchronize(pf->stream));
    if (pf->d_path) (void)hipFree(pf->d_path);
    pf->d_path = nullptr; pf->path_cap = 0;
    HIP_TRY(hipMalloc(&pf->d_path, need * 8));
    pf->path_cap = need;
  }
  HIP_TRY(hipMemsetAsync(pf->d_path, 0, need * 8, pf->stream));
  return CSSM_OK;
}
// row `row` of the path <- the current cloud's particle at GLOBAL slot `slot`, if this rank owns it
static void path_record(cssm_pf* pf, size_t row, uint64_t slot) {
  if (!pf->want_path || !pf->d_path || slot < pf->first || slot >= pf->first + pf->n) return;
  hipLaunchKernelGGL(k_pick_shard, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride, (const uint32_t*)(pf->anc_valid ? pf->anc : nullptr),
                     pf->anc_valid ? pf->src2 : (const double*)nullptr, pf->src2_stride, pf->n_split, slot - pf->first, pf->d, pf->d_path + row * (size_t)pf->d);
}
// after observation (record) s of the resident series has been propagated and, if weighted, resampled
static void path_after(cssm_pf* pf, size_t s) { if (pf->want_path) path_record(pf, s + 1, (uint64_t)pf->h_recs[s].pick); }

extern "C" int cssm_pf_shard_want_path(cssm_pf* pf, int on) {
  int rc = shard_check(pf);
  if (rc) return rc;
  pf->want_path = on != 0;
  return CSSM_OK;
}
extern "C" int cssm_pf_shard_get_path(cssm_pf* pf, double* out_host, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!out_host) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->want_path || !pf->d_path || (T + 1) * (size_t)pf->d > pf->path_cap) return fail(CSSM_ESTATE, "no path of %zu observations was recorded (cssm_pf_shard_want_path before the series)", T);
  rc = bounded_sync(pf);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(out_host, pf->d_path, (T + 1) * (size_t)pf->d * 8, hipMemcpyDeviceToHost));
  return CSSM_OK;
}
static int bounded_sync(cssm_pf* pf);
// record of the step propagated last: a ring of 64 for streaming steps, the whole series after shard_begin
static size_t last_rec_slot(const cssm_pf* pf) { return pf->series ? (size_t)(pf->step - 1 - pf->rec_base) : (size_t)((pf->step - 1) % 64); }

static int shard_check(cssm_pf* pf) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (!pf->sharded) return fail(CSSM_ESTATE, "handle was not created with cssm_pf_create_shard");
  HIP_TRY(hipSetDevice(pf->device));
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_init(cssm_pf* pf, double t0) {
  int rc = shard_check(pf);
  if (rc) return rc;
  pf->series = false;
  return cssm_launch_init(pf, t0);
}

extern "C" int cssm_pf_shard_propagate(cssm_pf* pf, double t, double obs, int has_obs, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!pf->initialised) return fail(CSSM_ESTATE, "shard_propagate before shard_init");
  if (pf->series) return fail(CSSM_ESTATE, "a series begun with shard_begin is stepped with shard_propagate_at");   // (before any record is touched)
  // one record slot per step, round-robin, so that an in-flight step never sees its record overwritten
  rc = cssm_ensure_recs(pf, 64);
  if (rc) return rc;
  const size_t slot = pf->step % 64;
  cssm_build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[slot]);
  rc = cssm_build_fsub(pf, slot, 1, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, slot, 1, true);
  if (rc) return rc;
  rc = shard_prepare_step(pf, pf->d_recs + slot, pf->h_recs[slot].has_obs, sums5_dev);
  if (rc) return rc;
  pf->t = t;
  pf->step++;
  return CSSM_OK;
}

// sums5_dev == nullptr (the library's own series loop): nobody gathers the sums again -- the single-collective exchange totals
// the unit sums in k_boundary_pack -- so ONE kernel does: k_tile_sums, which decodes the level from the gathered keys itself.
static int shard_sums_impl(cssm_pf* pf, const uint64_t* all_sums5_dev, int world, uint64_t* sums5_dev);
extern "C" int cssm_pf_shard_sums(cssm_pf* pf, const uint64_t* all_sums5_dev, int world, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!all_sums5_dev || !sums5_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  return shard_sums_impl(pf, all_sums5_dev, world, sums5_dev);
}
static int shard_sums_impl(cssm_pf* pf, const uint64_t* all_sums5_dev, int world, uint64_t* sums5_dev) {
  if (world < 1 || world > 64) return fail(CSSM_ESHARD, "world %d", world);
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits;
  if (!sums5_dev) {
    prof_begin(pf, CSSM_K_TILE_SUMS);
    hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc, pf->tileS, pf->tileS2, pf->ntiles,
                       pf->sup, pf->nunits, 0, -1, (const double*)nullptr, pf->d_logtab, (const StepRec*)(pf->d_recs + slot), 28u,
                       (const unsigned long long*)all_sums5_dev, world);
    prof_end(pf);
    HIP_TRY(hipGetLastError());
    pf->last_optimistic = false;
    pf->sums_ready = true;
    if (pf->series && pf->step > pf->rec_base && pf->snaps.size() >= pf->step - pf->rec_base) pf->snaps[pf->step - 1 - pf->rec_base].last_optimistic = false;
    return CSSM_OK;
  }
  prof_begin(pf, CSSM_K_TILE_SUMS);
  hipLaunchKernelGGL(k_import_level, dim3(1), dim3(1), 0, pf->stream, pf->sc, (const unsigned long long*)all_sums5_dev, world,
                     (const StepRec*)(pf->d_recs + slot));
  // (every kernel here returns at once while the series is on hold after a capacity miss or void: the level, the unit sums
  //  and the exported words of the observation it holds at are what cssm_pf_shard_resume's caller continues from)
  hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc, pf->tileS, pf->tileS2, pf->ntiles,
                     pf->sup, pf->nunits, 0, -1, (const double*)nullptr, pf->d_logtab, (const StepRec*)(pf->d_recs + slot), 28u);
  // word 4 (the max key) of sums5_dev is left as shard_propagate wrote it
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, pf->stream, pf->tileS, pf->tileS2, pf->tileP, pf->nunits, pf->sc, pf->n_global, 0,
                     (double*)nullptr, (int32_t*)nullptr, 0u, (const double*)nullptr, (unsigned long long*)sums5_dev, 0, 28u, 1);
  prof_end(pf);
  HIP_TRY(hipGetLastError());
  pf->last_optimistic = false;
  pf->sums_ready = true;
  // the snapshot a resume restores was taken right after the propagate: the sums it describes are these now
  if (pf->series && pf->step > pf->rec_base && pf->snaps.size() >= pf->step - pf->rec_base) pf->snaps[pf->step - 1 - pf->rec_base].last_optimistic = false;
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_offspring(cssm_pf* pf, const uint64_t* all_sums5_dev, int rank, int world,
                                       int64_t* send_first_dev, int64_t* send_count_dev, uint64_t* redo_flag_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!all_sums5_dev || !send_first_dev || !send_count_dev || !redo_flag_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d", rank, world);
  const uint64_t n_per = (pf->n_global + (uint64_t)world - 1) / (uint64_t)world;
  if (pf->first != (uint64_t)rank * n_per) return fail(CSSM_ESHARD, "rank %d must own particles from %llu (ceil(N/world) per rank), handle starts at %llu",
                                                       rank, (unsigned long long)((uint64_t)rank * n_per), (unsigned long long)pf->first);
  if (!pf->sums_ready) return fail(CSSM_ESTATE, "the exact exchange forms its sums with cssm_pf_shard_sums first (all-gather of the maxima, "
                                                "sums relative to the level they select, all-gather of the sums)");
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits;
  const int optimistic = pf->last_optimistic ? 1 : 0;
  // the rank's own particles write their runs inside the rank's slots straight into anc (indexed from the first
  // own slot); the end slots are kept for the send ranges; slots owned by other ranks' particles are filled by adopt
#define SHARD_OFF_ARGS pf->logw, pf->n, pf->sc, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, pf->d_recs + slot, pf->n_global, pf->endslot, \
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, optimistic ? 2 : 0, 0, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,            \
                     (const unsigned long long*)all_sums5_dev, rank, world, optimistic ? (int)pf->split : 1, pf->seed, (double*)nullptr, pf->d_logtab,    \
                     optimistic, (unsigned long long*)redo_flag_dev, (uint32_t)pf->first, (uint32_t)(pf->first + pf->n)
  if (pf->resampler == CSSM_RESAMPLE_STRATIFIED)
    hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_STRATIFIED>), dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, SHARD_OFF_ARGS);
  else
    hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_SYSTEMATIC>), dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, SHARD_OFF_ARGS);
#undef SHARD_OFF_ARGS
  pf->have_level = true;
  pf->send_first_dev = (const long long*)send_first_dev; pf->send_count_dev = (const long long*)send_count_dev;
  hipLaunchKernelGGL(k_send_ranges, dim3(1), dim3(64), 0, pf->stream, pf->endslot, pf->n, pf->sc, pf->d_recs + slot, pf->n_global, rank, world,
                     n_per, (long long*)send_first_dev, (long long*)send_count_dev, pf->resampler, pf->seed);
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}

// ---- a series known in advance: records resident on the device, observations propagated by index ----------------
// first j in [0, n) with endslot[j] > bound (strict) or >= bound, n if there is none; endslot is non-decreasing.
// All 64 lanes of a wave call it: every round probes 64 equally spaced positions of the bracket (4 rounds for 2^24).
__device__ __forceinline__ uint64_t wave_search_first(const uint32_t* __restrict__ endslot, uint64_t n, uint64_t bound, bool strict) {
  const int lane = threadIdx.x & 63;
  uint64_t lo = 0, hi = n;                                // the answer is in [lo, hi]
  while (hi > lo) {
    const uint64_t width = hi - lo;
    const uint64_t step = (width + 63) / 64;
    const uint64_t idx = lo + (uint64_t)lane * step;      // lane l probes the first element of its sub-range
    bool t = false;
    if (idx < hi) { const uint64_t v = endslot[idx]; t = strict ? (v > bound) : (v >= bound); }
    else t = true;                                        // beyond the bracket counts as "true" (hi itself is the fallback answer)
    const unsigned long long m = __ballot(t);
    const int f = m ? (__ffsll((long long)m) - 1) : 64;   // first lane whose probe is true
    if (f == 0) { hi = lo; break; }                       // the very first element of the bracket satisfies it
    // the probe of lane f-1 is false, the probe of lane f is true: the answer is in (idx_{f-1}, idx_f]
    const uint64_t new_lo = lo + (uint64_t)(f - 1) * step + 1;
    const uint64_t new_hi = (f < 64 && lo + (uint64_t)f * step < hi) ? lo + (uint64_t)f * step : hi;
    lo = new_lo; hi = new_hi;
    if (step == 1) { lo = hi = new_hi; break; }           // sub-ranges were single elements: idx_f (or hi) is the answer
  }
  return hi;
}

static int shard_prepare_step(cssm_pf* pf, const StepRec* d_rec, int weighted, uint64_t* sums5_dev) {
  // sums5_dev != nullptr: the level of the observation will come from the GLOBAL max (exact exchange; the "max" plan of a
  // series): only the local max is exported now and the sums are a pass of their own once the level is known
  // (cssm_pf_shard_sums), so the kernel that stores LOG-weights runs.  nullptr (single-collective exchange at every
  // observation's reference level): k_propagate<SUMS> forms the sums and stores the weights in place of the log-weights.
  // (LGCP, contract v8: the level is predicted from the weighted observation before -- the first event of a series has none and
  //  takes its level from the global max like an exact exchange: the caller must gather the maxima for it)
  if (weighted && sums5_dev == nullptr && pf->obs_kind == CSSM_OBS_LGCP && !pf->have_level)
    return fail(CSSM_ESTATE, "the first LGCP event of a series has no predicted level: its level comes from the all-gathered max (pass sums5_dev)");
  pf->safe_sums = sums5_dev != nullptr;
  int rc = cssm_launch_propagate(pf, d_rec);
  pf->safe_sums = false;
  if (rc) return rc;
  pf->sums_ready = pf->last_optimistic;
  if (weighted && sums5_dev) {   // (sums5_dev == nullptr: the single-collective exchange totals the sums in k_boundary_pack)
    // the rank's totals of the sub-unit sums k_propagate formed and the order key of its max -> 5 words for the all-gather
    const uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / pf->split;
    const uint32_t nsub = (uint32_t)((pf->n + chunk - 1) / chunk);
    // (LGCP -- !last_optimistic -- forms no sums in k_propagate: only the max travels, the sums read as zero.  On hold the
    //  kernel does nothing: the max slots were exported and cleared by the observation the series holds at.)
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, pf->stream, pf->tileS, pf->tileS2, pf->tileP, nsub, pf->sc, pf->n_global, 0,
                       (double*)nullptr, (int32_t*)nullptr, 0u, (const double*)nullptr, (unsigned long long*)sums5_dev, 1, 28u,
                       pf->last_optimistic ? 1 : 0);
    HIP_TRY(hipGetLastError());
  }
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_begin(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!t || !y) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data");
  rc = cssm_ensure_recs(pf, T);
  if (rc) return rc;
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];
  double tp = t0;
  for (size_t s = 0; s < T; ++s) { cssm_build_rec(pf, tp, t[s], y[s], has_obs ? has_obs[s] : 1, (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = cssm_build_fsub(pf, 0, T, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, 0, T, false);   // (a new cloud: no observation precedes the first record)
  if (rc) return rc;
  rc = cssm_launch_init(pf, t0);
  if (rc) return rc;
  pf->series = true;
  pf->rec_base = 0;
  rc = path_prepare(pf, T);
  if (rc) return rc;
  if (pf->want_path) {   // row 0: one particle of the initial cloud (:154)
    const int32_t pr = (int32_t)cssm_philox_draw(pf->seed, 0, 0, CSSM_STREAM_PICK, 0).v[0];
    const uint32_t pa = pr < 0 ? (uint32_t)0 - (uint32_t)pr : (uint32_t)pr;
    path_record(pf, 0, (uint64_t)pa % pf->n_global);
  }
  return CSSM_OK;
}

// T MORE observations of the sharded filter that is already running (what cssm_pf_ll_filter_more is to a single-GPU handle;
// Flow.scan(init)(stepFilter) handed the next T elements, model/ParticleFilter.scala:163-166): no new cloud, the first time
// increment from the handle's clock, record s of the call = observation (observations so far + s) of the filter (its Philox
// counter word).  Steps, exchanges, status and resume then work on the call's records exactly as after cssm_pf_shard_begin.
extern "C" int cssm_pf_shard_continue(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!t || !y) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data");
  if (!pf->initialised) return fail(CSSM_ESTATE, "cssm_pf_shard_continue before cssm_pf_shard_begin / _init");
  rc = bounded_sync(pf);   // (the records of the series before are rewritten)
  if (rc) return rc;
  rc = cssm_ensure_recs(pf, T);
  if (rc) return rc;
  double tp = pf->t;
  for (size_t s = 0; s < T; ++s) { cssm_build_rec(pf, tp, t[s], y[s], has_obs ? has_obs[s] : 1, pf->step + (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = cssm_build_fsub(pf, 0, T, true);
  if (rc) return rc;
  rc = cssm_upload_recs(pf, 0, T, true);
  if (rc) return rc;
  pf->series = true;
  pf->rec_base = pf->step;
  pf->snaps.clear();
  return path_prepare(pf, T);   // (row 0 of a continued call's path stays zero: the state before its first observation belongs to the call before)
}

extern "C" int cssm_pf_shard_propagate_at(cssm_pf* pf, size_t s, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!pf->series || !pf->initialised) return fail(CSSM_ESTATE, "shard_propagate_at before shard_begin");
  if (s >= pf->h_recs_cap || s + pf->rec_base != pf->step) return fail(CSSM_ESTATE, "steps of a series run in order (expected %u)", pf->step - pf->rec_base);
  if (pf->pre_snaps.size() <= s) pf->pre_snaps.resize(s + 1);
  pf->pre_snaps[s] = {pf->cur, pf->src, pf->src_stride, pf->src2, pf->src2_stride, pf->n_split, pf->anc_valid, pf->last_optimistic, pf->step, pf->t, pf->have_level, pf->wparity, pf->last_grp};
  rc = shard_prepare_step(pf, pf->d_recs + s, pf->h_recs[s].has_obs, sums5_dev);
  if (rc) return rc;
  pf->step++;
  pf->t = pf->h_recs[s].t_obs;
  if (!pf->h_recs[s].has_obs) path_after(pf, s);   // (a weighted observation: behind its resampling, cssm_pf_shard_adopt_spec / _adopt)
  if (pf->snaps.size() <= s) pf->snaps.resize(s + 1);
  pf->snaps[s] = {pf->cur, pf->src, pf->src_stride, pf->src2, pf->src2_stride, pf->n_split, pf->anc_valid, pf->last_optimistic, pf->step, pf->t, pf->have_level, pf->wparity, pf->last_grp};
  return CSSM_OK;
}

// A capacity miss of the single-collective series is not the end of the series.  k_offspring_expand_spec of the observation
// that missed did nothing (on every rank alike: the verdict is a function of the segment headers), recorded the observation
// index, and every later kernel returned at once.  This call reads that index, clears the bit and rewinds the host-side
// state to "observation fail_step propagated, not yet resampled": the host then redoes that observation's exchange with
// a larger capacity (boundary_pack, all-to-all, adopt_spec) and continues the series after it.
extern "C" int cssm_pf_shard_resume(cssm_pf* pf, uint32_t* fail_step_out) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!fail_step_out) return fail(CSSM_EINVAL_ARG, "null argument");
  rc = bounded_sync(pf);   // (first: a copy into pageable memory would wait for the stream without a bound)
  if (rc) return rc;
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (!(h.err & 8u) || h.fail_step == 0xffffffffu || h.fail_step < pf->rec_base || h.fail_step - pf->rec_base >= pf->snaps.size())
    return fail(CSSM_ESTATE, "no resumable capacity miss is recorded");
  if (h.err & 7u) return fail(CSSM_ESTATE, "the series has other errors (bits %u)", h.err);
  const uint32_t s = h.fail_step - pf->rec_base;      // (the record's index in the resident series)
  h.err &= ~8u; h.fail_step = 0xffffffffu;
  // only err and fail_step change on the device (ll, ess, sums stay what the last completed observation left)
  HIP_TRY(hipMemcpyAsync(&pf->sc->err, &h.err, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &h.fail_step, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  // (pack blocks of the held launches that returned without their ticket while siblings took theirs would leave a count behind)
  if (pf->peer_tickets) {
    HIP_TRY(hipMemsetAsync(pf->peer_tickets, 0, 96 * sizeof(unsigned int), pf->stream));
    HIP_TRY(hipMemsetAsync(pf->peer_tickets + CSSM_PEER_TICKET_NEED, 0, 128 * sizeof(unsigned int), pf->stream));   // (PackNeed::need and the second set of tickets likewise)
  }
  HIP_TRY(hipStreamSynchronize(pf->stream));
  const cssm_pf::Snap& q = pf->snaps[s];
  pf->cur = q.cur; pf->src = q.src; pf->src_stride = q.src_stride; pf->src2 = q.src2; pf->src2_stride = q.src2_stride;
  pf->n_split = q.n_split; pf->anc_valid = q.anc_valid; pf->last_optimistic = q.last_optimistic; pf->step = q.step; pf->t = q.t;
  pf->have_level = q.have_level; pf->wparity = q.wparity; pf->last_grp = q.last_grp;
  pf->sums_ready = true;   // (the exchange that missed ran behind them)
  *fail_step_out = s;
  return CSSM_OK;
}

// An observation whose reference level its max rules out (sticky bit 4) is not the end of the series either: the launch that found
// out resampled nothing on any rank (the verdict is a function of the segment headers), recorded the observation and every later
// kernel returned at once.  This call rewinds the host-side state to "observation fail_step NOT yet propagated": its source cloud, the
// ancestors and received rows of the observation before are untouched (the held propagate wrote the OTHER state buffer; the failed
// exchange the other receive window).  The host propagates it again storing log-weights, takes its level from the all-gathered max
// (cssm_pf_shard_propagate_at with sums5_dev, cssm_pf_shard_sums) and continues the series behind it on the ordinary plan.
extern "C" int cssm_pf_shard_resume_level(cssm_pf* pf, uint32_t* fail_step_out) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!fail_step_out) return fail(CSSM_EINVAL_ARG, "null argument");
  rc = bounded_sync(pf);
  if (rc) return rc;
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (!(h.err & 4u) || h.fail_step == 0xffffffffu || h.fail_step < pf->rec_base || h.fail_step - pf->rec_base >= pf->pre_snaps.size())
    return fail(CSSM_ESTATE, "no observation with a ruled-out reference level is recorded");
  if (h.err & ~4u) return fail(CSSM_ESTATE, "the series has other errors (bits %u)", h.err);
  const uint32_t s = h.fail_step - pf->rec_base;
  h.err &= ~4u; h.fail_step = 0xffffffffu;
  HIP_TRY(hipMemcpyAsync(&pf->sc->err, &h.err, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &h.fail_step, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  if (pf->peer_tickets) {
    HIP_TRY(hipMemsetAsync(pf->peer_tickets, 0, 96 * sizeof(unsigned int), pf->stream));
    HIP_TRY(hipMemsetAsync(pf->peer_tickets + CSSM_PEER_TICKET_NEED, 0, 128 * sizeof(unsigned int), pf->stream));   // (PackNeed::need and the second set of tickets likewise)
  }
  HIP_TRY(hipStreamSynchronize(pf->stream));
  const cssm_pf::Snap& q = pf->pre_snaps[s];
  pf->cur = q.cur; pf->src = q.src; pf->src_stride = q.src_stride; pf->src2 = q.src2; pf->src2_stride = q.src2_stride;
  pf->n_split = q.n_split; pf->anc_valid = q.anc_valid; pf->last_optimistic = q.last_optimistic; pf->step = q.step; pf->t = q.t;
  pf->have_level = q.have_level; pf->wparity = q.wparity; pf->last_grp = q.last_grp;
  pf->sums_ready = false;
  *fail_step_out = s;
  return CSSM_OK;
}

// ---- single-collective exchange (k_boundary_pack / k_expand_spec in cssm_kernels.hip.h)

extern "C" int64_t cssm_pf_shard_unit(const cssm_pf* pf) {
  return pf ? (int64_t)((uint64_t)pf->sup * CSSM_TILE / pf->split) : 0;
}

extern "C" int64_t cssm_pf_shard_spec_segment(const cssm_pf* pf, int64_t cap) {
  return (pf && cap >= 1) ? (int64_t)spec_seg(pf->d, (long long)cap) : 0;
}

static int boundary_pack_impl(cssm_pf* pf, int rank, int world, int64_t cap, double* send_buf_dev, bool peer, int phase = 7);
// rows next to the boundary that the peer-written exchange writes at once (>= cap: every row travels)
static long long peer_eager_rows(const cssm_pf* pf, int64_t cap) { return pf->peer_all_rows ? (long long)cap : std::min<long long>(pf->peer_eager, (long long)cap); }
extern "C" int cssm_pf_shard_boundary_pack(cssm_pf* pf, int rank, int world, int64_t cap, double* send_buf_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!send_buf_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  return boundary_pack_impl(pf, rank, world, cap, send_buf_dev, false);
}
static int boundary_pack_impl(cssm_pf* pf, int rank, int world, int64_t cap, double* send_buf_dev, bool peer, int phase) {
  // phase (peer): bit 0 = headers and unit-sum prefixes, bit 1 = eager rows, bit 2 = needed rows beyond them (boundary_pack_block)
  if (cap < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d / cap %lld", rank, world, (long long)cap);
  // !last_optimistic: the sums were formed by cssm_pf_shard_sums relative to the level chosen with the all-gathered max
  const size_t slot = last_rec_slot(pf);
  // (the sums a propagate formed are per block -- pf->split blocks per unit; the sums of cssm_pf_shard_sums per unit)
  const uint32_t split = pf->last_optimistic ? pf->split : 1u;
  const uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / split;
  const uint32_t nsub = (uint32_t)((pf->n + chunk - 1) / chunk);
  const long long cnt = std::min<long long>((long long)pf->n, (long long)cap);
  const int tiles = (int)((cnt + CSSM_TILE - 1) / CSSM_TILE);
  // group sums at hand (the propagate's blocks accumulated them in set pf->wparity): the header blocks total those, and the offspring
  // blocks of the launch behind this one take their prefixes from them -- no prefix block
  const int grp_set = (pf->last_grp && pf->last_optimistic) ? pf->wparity : -1;
  const bool pre = nsub <= 8u * CSSM_BLOCK && grp_set < 0;   // (the prefix block's reach)
  prof_begin(pf, CSSM_K_PACK);
  hipLaunchKernelGGL(k_boundary_pack, dim3(tiles + 2, world), dim3(CSSM_BLOCK), 0, pf->stream, pf->state[pf->cur], pf->stride, pf->logw, pf->n, pf->d,
                     world, rank, (long long)cap, pf->d_recs + slot, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, nsub,
                     (const Scalars*)pf->sc, send_buf_dev, chunk, pf->last_optimistic ? 0 : 1,
                     pre ? pf->unitPre : (cssm_u128*)nullptr,
                     peer ? (const PeerTable*)pf->peer_tab : (const PeerTable*)nullptr, (int)(pf->peer_seq & 1u), pf->peer_seq, pf->peer_tickets, grp_set,
                     phase, peer ? peer_eager_rows(pf, cap) : (long long)cap, pf->n_global, pf->seed, pf->resampler, (uint64_t)pf->first, (uint64_t)(pf->first + pf->n));
  pf->spec_pre = pre;   // (k_offspring_expand_spec reads them: cssm_pf_shard_adopt_spec)
  prof_end(pf);
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}

static int adopt_spec_impl(cssm_pf* pf, const double* recv_buf_dev, int rank, int world, int64_t cap, const unsigned int* peer_flags, uint32_t peer_seq,
                           bool merged = false);
extern "C" int cssm_pf_shard_adopt_spec(cssm_pf* pf, const double* recv_buf_dev, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!recv_buf_dev) return fail(CSSM_EINVAL_ARG, "recv_buf_dev is null");
  return adopt_spec_impl(pf, recv_buf_dev, rank, world, cap, nullptr, 0u);
}
static int adopt_spec_impl(cssm_pf* pf, const double* recv_buf_dev, int rank, int world, int64_t cap, const unsigned int* peer_flags, uint32_t peer_seq,
                           bool merged) {
  // merged (peer-written exchange, one shard per stream): the pack blocks ride at the head of the same launch (k_exchange_offspring)
  if (cap < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d / cap %lld", rank, world, (long long)cap);
  const uint64_t n_per = (pf->n_global + (uint64_t)world - 1) / (uint64_t)world;
  if (pf->first != (uint64_t)rank * n_per) return fail(CSSM_ESHARD, "rank %d must own particles from %llu", rank, (unsigned long long)((uint64_t)rank * n_per));
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits + CSSM_SPEC_EXPAND_BLOCKS;   // (the blocks that expand the received rows lead the offspring blocks)
  const long long seg = spec_seg(pf->d, (long long)cap);
  // the 5 words of every rank are the header words 1..5 of its segment (the all-to-all delivered this rank's own too)
  const unsigned long long* all5 = reinterpret_cast<const unsigned long long*>(recv_buf_dev) + 1;
  const uint32_t n_split = (uint32_t)pf->n;
  // the set of group sums of this observation (every exchange of the handle rotates it, whether the sums were used or not: block 0 of
  // the offspring blocks clears the two other sets); grp: the propagate's blocks accumulated them -- the GRP instantiations
  const int gset = pf->wparity;
  const bool grp = pf->last_grp && pf->last_optimistic;
  prof_begin(pf, CSSM_K_EXPAND);
  if (merged) {
    if (!pf->last_optimistic) return fail(CSSM_ESTATE, "the merged peer exchange serves the propagate that formed the sums (levels known in advance)");
    const uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / pf->split;
    const uint32_t nsub = (uint32_t)((pf->n + chunk - 1) / chunk);
    const long long cnt = std::min<long long>((long long)pf->n, (long long)cap);
    const bool pre = nsub <= 8u * CSSM_BLOCK && !grp;
    PackArgs pk;
    pk.src = pf->state[pf->cur]; pk.stride = pf->stride; pk.nsub = nsub; pk.chunk = chunk;
    pk.pre_out = pre ? pf->unitPre : (cssm_u128*)nullptr;
    pk.peer = (const PeerTable*)pf->peer_tab; pk.parity = (int)(peer_seq & 1u); pk.tickets = pf->peer_tickets;
    pk.pre_flag = pre ? pf->peer_tickets + 64 : (unsigned int*)nullptr;
    pk.pack_gx = (uint32_t)((cnt + CSSM_TILE - 1) / CSSM_TILE) + 2u;
    pk.eager = peer_eager_rows(pf, cap);
    pf->spec_pre = pre;
    auto kx = (pf->resampler == CSSM_RESAMPLE_STRATIFIED)
                  ? (grp ? k_exchange_offspring<2, CSSM_RESAMPLE_STRATIFIED, true> : k_exchange_offspring<2, CSSM_RESAMPLE_STRATIFIED>)
                  : (grp ? k_exchange_offspring<2, CSSM_RESAMPLE_SYSTEMATIC, true> : k_exchange_offspring<2, CSSM_RESAMPLE_SYSTEMATIC>);
    hipLaunchKernelGGL(kx, dim3(tgrid + (int)(pk.pack_gx * (uint32_t)world)), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc,
                     (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, (const StepRec*)(pf->d_recs + slot), pf->n_global, pf->endslot,
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, 2, gset, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,
                     all5, rank, world, (int)pf->split, pf->seed, (double*)nullptr, (const double*)pf->d_logtab,
                     2, (unsigned long long*)(pf->d_xch + 128), (uint32_t)pf->first, (uint32_t)(pf->first + pf->n), (uint32_t)seg,
                     recv_buf_dev, (long long)cap, pf->d, n_split, pre ? (const cssm_u128*)pf->unitPre : (const cssm_u128*)nullptr,
                     peer_flags, peer_seq, pk);
  } else if (pf->last_optimistic) {
    auto ke = (pf->resampler == CSSM_RESAMPLE_STRATIFIED)
                  ? (grp ? k_offspring_expand_spec<2, CSSM_RESAMPLE_STRATIFIED, true> : k_offspring_expand_spec<2, CSSM_RESAMPLE_STRATIFIED>)
                  : (grp ? k_offspring_expand_spec<2, CSSM_RESAMPLE_SYSTEMATIC, true> : k_offspring_expand_spec<2, CSSM_RESAMPLE_SYSTEMATIC>);
    hipLaunchKernelGGL(ke, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc,
                     (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, (const StepRec*)(pf->d_recs + slot), pf->n_global, pf->endslot,
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, pf->last_optimistic ? 2 : 0, gset, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,
                     all5, rank, world, pf->last_optimistic ? (int)pf->split : 1, pf->seed, (double*)nullptr, (const double*)pf->d_logtab,
                     pf->last_optimistic ? 2 : 0, (unsigned long long*)(pf->d_xch + 128), (uint32_t)pf->first, (uint32_t)(pf->first + pf->n), (uint32_t)seg,
                     recv_buf_dev, (long long)cap, pf->d, n_split, pf->spec_pre ? (const cssm_u128*)pf->unitPre : (const cssm_u128*)nullptr,
                     peer_flags, peer_seq, peer_flags ? peer_eager_rows(pf, cap) : (long long)cap);
  } else {
    auto ke = (pf->resampler == CSSM_RESAMPLE_STRATIFIED) ? k_offspring_expand_spec<0, CSSM_RESAMPLE_STRATIFIED> : k_offspring_expand_spec<0, CSSM_RESAMPLE_SYSTEMATIC>;
    hipLaunchKernelGGL(ke, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc,
                     (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, (const StepRec*)(pf->d_recs + slot), pf->n_global, pf->endslot,
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, pf->last_optimistic ? 2 : 0, gset, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,
                     all5, rank, world, pf->last_optimistic ? (int)pf->split : 1, pf->seed, (double*)nullptr, (const double*)pf->d_logtab,
                     pf->last_optimistic ? 2 : 0, (unsigned long long*)(pf->d_xch + 128), (uint32_t)pf->first, (uint32_t)(pf->first + pf->n), (uint32_t)seg,
                     recv_buf_dev, (long long)cap, pf->d, n_split, pf->spec_pre ? (const cssm_u128*)pf->unitPre : (const cssm_u128*)nullptr,
                     peer_flags, peer_seq, peer_flags ? peer_eager_rows(pf, cap) : (long long)cap);
  }
  prof_end(pf);
  HIP_TRY(hipGetLastError());
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride;
  pf->src2 = recv_buf_dev; pf->src2_stride = 0; pf->n_split = n_split; pf->anc_valid = true;   // stride 0 = rows of d + 1
  pf->wmode = pf->last_optimistic;
  pf->have_level = true;        // (its block 0 published the next observation's predicted level: LGCP)
  pf->wparity = (pf->wparity + 1) % CSSM_MAXSETS;
  if (pf->series) path_after(pf, slot);   // (on hold after a capacity miss the slot index of a wrong row is overwritten when the observation is redone)
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_wait_stats(cssm_pf* pf, uint64_t* out4) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!out4) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->h_sc) return fail(CSSM_ESTATE, "no status has been read");
  out4[0] = pf->h_sc->xwaits; out4[1] = pf->h_sc->hdr_wait_ticks; out4[2] = pf->h_sc->xhdr_wait_ticks; out4[3] = pf->h_sc->rows_wait_ticks;
  return CSSM_OK;
}

// ll, ess and the sticky bits of a series run with the single-collective exchange: bit 2 (value 4) = some observation's
// reference level was ruled out by the max, bit 3 (value 8) = a capacity miss that was not resumed.  Either bit means the
// numbers are not the filter's (ShardedFilter moves on to its next plan); `need` (optional, T entries): diagnostics.
extern "C" int cssm_pf_shard_status(cssm_pf* pf, double* ll_out, int32_t* ess_out, uint32_t* bits_out, uint32_t* need, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  // the scalars the host reads travel as at the end of a single-GPU call: k_finish writes them into host-mapped memory behind the
  // series, the one (bounded) wait for the stream is the only synchronisation -- no device-to-host copy, no second wait (round 3:
  // two waits and two copies per call, ~100 us of a 20-observation leg)
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(CSSM_BLOCK), 0, pf->stream, pf->sc, (const cssm_u128*)pf->s2buf, pf->s2_stride, (double*)nullptr, (int32_t*)nullptr, 0u,
                     pf->gen, pf->hd_sc, (double*)nullptr, (int32_t*)nullptr, (uint32_t*)nullptr, 0u);
  HIP_TRY(hipGetLastError());
  rc = bounded_sync(pf);
  if (rc) return rc;
  Scalars h;                    // (the mirror is valid from `err` on)
  memcpy(reinterpret_cast<char*>(&h) + CSSM_SC_TAIL_OFF, reinterpret_cast<const char*>(pf->h_sc) + CSSM_SC_TAIL_OFF, sizeof(Scalars) - CSSM_SC_TAIL_OFF);
  if (need) memset(need, 0, T * 4);   // (diagnostics of the removed two-collective exchange: nothing records them any more)
  prof_collect(pf);
  if (ll_out) *ll_out = h.ll;
  if (ess_out) *ess_out = h.ess;
  if (bits_out) *bits_out = h.err & 12u;
  h.err &= ~12u;
  if (h.err & 16u) return fail(CSSM_ESHARD, "peer-written exchange: a rank's segment of observation %u did not arrive within the wait bound "
                                            "(a peer that died, or never enqueued its series?  wait code %#x: bit 0 / 1 / 2 header words awaited by an offspring / "
                                            "expansion / pack row block, 3 the eager rows' flag, 4 the flag of the rows beyond them, 5 a header flag; bits 8-15 the rank)",
                                            h.fail_step, h.wait_code);
  return cssm_check_device_err(pf, h);
}

// ------------------------------------------------------------------------------------ peer-written exchange
// (kernel side and protocol: PeerTable in cssm_shard_kernels.hip.h.)  Every rank allocates ONE slab -- two receive windows of
// `world` segments each, then two sets of flags -- and hands out a cssm_peer_handle for it; cssm_pf_shard_peer_connect maps the
// other ranks' slabs (hipIpcOpenMemHandle; the plain pointer where the owner lives in this process: several shards of one
// process, or the rank itself) and fills the device table the pack kernel reads.
struct PeerState {
  int world = 0, rank = 0;
  int64_t cap = 0;
  size_t seg = 0, win_bytes = 0, slab_bytes = 0;
  void* slab = nullptr;                 // own: [window 0 | window 1 | flags 0 | flags 1]
  bool slab_fine = false;
  void* mapped[64] = {nullptr};         // peers' slabs as this process sees them
  bool opened[64] = {false};            // ... through hipIpcOpenMemHandle (to be closed)
  bool connected = false;
};
static size_t peer_flags_bytes(int world) { return (size_t)world * CSSM_PEER_FLAG_STRIDE * sizeof(unsigned int); }
static double* peer_window(const PeerState* ps, void* slab, int p) { return reinterpret_cast<double*>(static_cast<char*>(slab) + (size_t)p * ps->win_bytes); }
static unsigned int* peer_flagset(const PeerState* ps, void* slab, int p) {
  return reinterpret_cast<unsigned int*>(static_cast<char*>(slab) + 2 * ps->win_bytes + (size_t)p * peer_flags_bytes(ps->world));
}
void cssm_peer_free(cssm_pf* pf) {
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps) return;
  for (int q = 0; q < 64; ++q) if (ps->opened[q] && ps->mapped[q]) (void)hipIpcCloseMemHandle(ps->mapped[q]);
  if (ps->slab) (void)hipFree(ps->slab);
  if (pf->peer_tab) (void)hipFree(pf->peer_tab);
  if (pf->peer_tickets) (void)hipFree(pf->peer_tickets);
  pf->peer_tab = nullptr; pf->peer_tickets = nullptr;
  delete ps;
  pf->peer = nullptr;
}

extern "C" int cssm_pf_shard_peer_setup(cssm_pf* pf, int rank, int world, int64_t cap, cssm_peer_handle* mine_out) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!mine_out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (cap < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d / cap %lld", rank, world, (long long)cap);
  rc = bounded_sync(pf);   // (windows of an earlier setup may still be gathered from)
  if (rc) return rc;
  if (pf->src2 && pf->peer && static_cast<PeerState*>(pf->peer)->slab) {
    // the current cloud may still point into the old windows (ancestors of the last exchange): materialise nothing, refuse
    const PeerState* old = static_cast<PeerState*>(pf->peer);
    const char* a = static_cast<const char*>(old->slab);
    if (reinterpret_cast<const char*>(pf->src2) >= a && reinterpret_cast<const char*>(pf->src2) < a + old->slab_bytes)
      return fail(CSSM_ESTATE, "the cloud still reads rows of the current peer windows: set the windows up before the series (or with the same capacity)");
  }
  cssm_peer_free(pf);
  PeerState* ps = new PeerState();
  pf->peer = ps;
  ps->world = world; ps->rank = rank; ps->cap = cap;
  ps->seg = (size_t)spec_seg(pf->d, (long long)cap);
  ps->win_bytes = ((size_t)world * ps->seg * sizeof(double) + 255) / 256 * 256;
  ps->slab_bytes = 2 * ps->win_bytes + 2 * peer_flags_bytes(world);
  // fine-grained device memory (visible to peers while a kernel runs; not held in this GPU's L2 across the peers' writes);
  // plain device memory if the runtime refuses (single-GPU use is indifferent)
  static const bool coarse = getenv("CSSM_PEER_COARSE") != nullptr;
  if (!coarse && hipExtMallocWithFlags(&ps->slab, ps->slab_bytes, hipDeviceMallocFinegrained) == hipSuccess) ps->slab_fine = true;
  else { (void)hipGetLastError(); if (hipMalloc(&ps->slab, ps->slab_bytes) != hipSuccess) { ps->slab = nullptr; cssm_peer_free(pf); return fail(CSSM_ENOMEM, "peer windows (%zu bytes)", ps->slab_bytes); } }
  HIP_TRY(hipMemsetAsync(ps->slab, 0, ps->slab_bytes, pf->stream));
  HIP_TRY(hipMalloc(&pf->peer_tab, sizeof(PeerTable)));
  HIP_TRY(hipMalloc(&pf->peer_tickets, CSSM_PEER_TICKET_WORDS * sizeof(unsigned int)));   // (layout: CSSM_PEER_TICKET_WORDS in cssm_shard_kernels.hip.h)
  HIP_TRY(hipMemsetAsync(pf->peer_tickets, 0, CSSM_PEER_TICKET_WORDS * sizeof(unsigned int), pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  memset(mine_out, 0, sizeof *mine_out);
  mine_out->pid = (uint64_t)getpid();
  mine_out->local_ptr = (uint64_t)(uintptr_t)ps->slab;
  mine_out->bytes = (uint64_t)ps->slab_bytes;
  mine_out->device = pf->device;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, ps->slab) == hipSuccess) { memcpy(mine_out->ipc, &h, sizeof h); mine_out->has_ipc = 1; }
  else { (void)hipGetLastError(); mine_out->has_ipc = 0; }   // (usable inside this process only)
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_peer_connect(cssm_pf* pf, const cssm_peer_handle* all, int world) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->slab) return fail(CSSM_ESTATE, "cssm_pf_shard_peer_connect before cssm_pf_shard_peer_setup");
  if (!all || world != ps->world) return fail(CSSM_ESHARD, "peer handles of %d ranks expected", ps->world);
  const uint64_t me = (uint64_t)getpid();
  PeerTable tab;
  memset(&tab, 0, sizeof tab);
  for (int q = 0; q < world; ++q) {
    if (all[q].bytes != (uint64_t)ps->slab_bytes) return fail(CSSM_ESHARD, "rank %d set its windows up for another capacity or world (%llu bytes, %zu here)", q, (unsigned long long)all[q].bytes, ps->slab_bytes);
    void* base = nullptr;
    // windows that another GPU or another process writes while kernels of this one run must be fine-grained memory: without it nothing
    // says when the writes become visible here.  No silent use of plain device memory (CSSM_PEER_COARSE=1 asks for it by name: A/B on one GPU)
    if (q != ps->rank && !ps->slab_fine && (all[q].pid != me || all[q].device != pf->device) && getenv("CSSM_PEER_COARSE") == nullptr)
      return fail(CSSM_ESHARD, "this rank's receive windows are not fine-grained device memory (hipExtMallocWithFlags refused) and rank %d writes them from "
                               "another %s: the peer-written exchange is refused", q, all[q].pid != me ? "process" : "device");
    if (q == ps->rank) base = ps->slab;
    else if (all[q].pid == me) base = (void*)(uintptr_t)all[q].local_ptr;       // a shard of this process
    else {
      if (!all[q].has_ipc) return fail(CSSM_ESHARD, "rank %d exported no IPC handle for its windows", q);
      hipIpcMemHandle_t h;
      memcpy(&h, all[q].ipc, sizeof h);
      const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { (void)hipGetLastError(); return fail(CSSM_EHIP, "hipIpcOpenMemHandle of rank %d's windows: %s", q, hipGetErrorString(e)); }
      ps->opened[q] = true;
    }
    ps->mapped[q] = base;
    for (int p = 0; p < 2; ++p) { tab.win[p][q] = peer_window(ps, base, p); tab.flag[p][q] = peer_flagset(ps, base, p); }
  }
  HIP_TRY(hipMemcpy(pf->peer_tab, &tab, sizeof tab, hipMemcpyHostToDevice));
  ps->connected = true;
  return CSSM_OK;
}

// One empty round of the protocol across all ranks (every rank calls it at the same point, e.g. behind a barrier of the host's
// control plane): CSSM_OK, or CSSM_ESHARD naming how many ranks' tokens did not arrive within the wait bound.
extern "C" int cssm_pf_shard_peer_handshake(cssm_pf* pf, uint32_t token) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->connected) return fail(CSSM_ESTATE, "peer windows are not set up (cssm_pf_shard_peer_setup / _connect)");
  if (token == 0u) return fail(CSSM_EINVAL_ARG, "the token must not be zero (the flags start there)");
  if (ps->seg < (size_t)CSSM_PEER_PROBE_WORDS) return fail(CSSM_ESHARD, "peer handshake: segments of %zu doubles are too small for the probe", ps->seg);
  unsigned int* res = pf->peer_tickets + 96;   // (words of the ticket allocation nobody else uses: [0] missing tokens, [1] / [2] probe words read wrongly)
  HIP_TRY(hipMemsetAsync(res, 0, 4 * sizeof(unsigned int), pf->stream));
  hipLaunchKernelGGL(k_peer_handshake, dim3(1), dim3(64), 0, pf->stream, (const PeerTable*)pf->peer_tab, ps->world, ps->rank, token, res, pf->peer_wait_ticks, ps->seg);
  // ... and, behind a kernel boundary as the propagate is behind the exchange: what the peers' handshake kernels wrote into this rank's window
  hipLaunchKernelGGL(k_peer_verify, dim3(1), dim3(CSSM_BLOCK), 0, pf->stream, (const double*)peer_window(ps, ps->slab, 0), ps->world, token, ps->seg, res);
  HIP_TRY(hipGetLastError());
  unsigned int got[4] = {0u, 0u, 0u, 0u};
  HIP_TRY(hipMemcpyAsync(got, res, sizeof got, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  pf->peer_probe_plain_bad = got[2];
  if (got[0]) return fail(CSSM_ESHARD, "peer handshake: the tokens of %u of %d ranks did not arrive in this rank's windows within the wait bound", got[0], ps->world);
  if (got[1]) return fail(CSSM_ESHARD, "peer handshake: %u probe words the peers wrote into this rank's window read wrongly (system-scope loads behind the flag): "
                                        "peer-written windows are not coherent on this system", got[1]);
  return CSSM_OK;
}
// probe words of the last handshake that PLAIN loads of the window read wrongly while system-scope loads read them right (diagnostic: the
// library reads windows with system-scope loads only)
extern "C" uint32_t cssm_pf_shard_peer_probe_stale(const cssm_pf* pf) { return pf ? pf->peer_probe_plain_bad : 0u; }

extern "C" void cssm_pf_shard_peer_close(cssm_pf* pf) {
  if (!pf) return;
  (void)hipSetDevice(pf->device);
  if (pf->stream) (void)hipStreamSynchronize(pf->stream);
  if (pf->peer && pf->src2) {   // the cloud must not keep pointing into windows that are about to go
    const PeerState* ps = static_cast<PeerState*>(pf->peer);
    const char* a = static_cast<const char*>(ps->slab);
    if (a && reinterpret_cast<const char*>(pf->src2) >= a && reinterpret_cast<const char*>(pf->src2) < a + ps->slab_bytes) pf->initialised = false;
  }
  cssm_peer_free(pf);
}

// Stage calls of the peer-written exchange (the library's own loop below, and a host that drives several shards of one process):
// cssm_pf_shard_pack_peer = k_boundary_pack writing into the peers' windows + flags; cssm_pf_shard_adopt_peer = k_offspring_expand_spec
// on this rank's window of the same exchange, behind the flags.  Both use the handle's exchange counter: one pack, then one
// adopt, per weighted observation, on every rank alike.
extern "C" int cssm_pf_shard_pack_peer(cssm_pf* pf, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->connected) return fail(CSSM_ESTATE, "peer windows are not set up (cssm_pf_shard_peer_setup / _connect)");
  if (rank != ps->rank || world != ps->world || cap != ps->cap) return fail(CSSM_ESHARD, "peer windows were set up for rank %d / world %d / cap %lld", ps->rank, ps->world, (long long)ps->cap);
  if (pf->peer_packed) return fail(CSSM_ESTATE, "cssm_pf_shard_pack_peer twice without cssm_pf_shard_adopt_peer");
  pf->peer_seq++;
  pf->peer_packed = true;
  const bool all = peer_eager_rows(pf, cap) >= (long long)cap;
  pf->peer_rows_packed = all;
  // (the rows beyond the eager ones are a stage of their own, cssm_pf_shard_pack_rows_peer -- their blocks wait for every rank's header)
  return boundary_pack_impl(pf, rank, world, cap, nullptr, true, all ? 7 : 3);
}
// Second stage of the pack: the rows of the boundary blocks that the neighbours' slots need, once every rank's header is on its way --
// a host that drives several shards on ONE stream calls cssm_pf_shard_pack_peer on all of them, then this on all of them, then
// cssm_pf_shard_adopt_peer on all of them.  (CSSM_PEER_ALL_ROWS=1: the first stage wrote every row; nothing left to do here.)
extern "C" int cssm_pf_shard_pack_rows_peer(cssm_pf* pf, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->connected) return fail(CSSM_ESTATE, "peer windows are not set up (cssm_pf_shard_peer_setup / _connect)");
  if (rank != ps->rank || world != ps->world || cap != ps->cap) return fail(CSSM_ESHARD, "peer windows were set up for rank %d / world %d / cap %lld", ps->rank, ps->world, (long long)ps->cap);
  if (!pf->peer_packed) return fail(CSSM_ESTATE, "cssm_pf_shard_pack_rows_peer without cssm_pf_shard_pack_peer");
  if (pf->peer_rows_packed) return CSSM_OK;
  pf->peer_rows_packed = true;
  return boundary_pack_impl(pf, rank, world, cap, nullptr, true, 4);
}
// Diagnostics: this rank's flag words of both windows (world x CSSM_PEER_FLAG_STRIDE uint32 each, window 0 first), its local ticket words
// behind them (CSSM_PEER_TICKET_WORDS) and, last, the handle's exchange counter -- as they are now (no synchronisation with the stream: for
// a series that has just ended in "a rank's segment did not arrive").  Returns the number of words written, or a negative error.
extern "C" int64_t cssm_pf_shard_peer_debug(cssm_pf* pf, uint32_t* out, size_t nwords) {
  if (!pf || !out) return CSSM_EINVAL_ARG;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->slab) return CSSM_ESTATE;
  const size_t fw = (size_t)ps->world * CSSM_PEER_FLAG_STRIDE, need = 2 * fw + CSSM_PEER_TICKET_WORDS + 1;
  if (nwords < need) return CSSM_EINVAL_ARG;
  for (int p = 0; p < 2; ++p)
    if (hipMemcpy(out + (size_t)p * fw, peer_flagset(ps, ps->slab, p), fw * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return CSSM_EHIP; }
  if (hipMemcpy(out + 2 * fw, pf->peer_tickets, CSSM_PEER_TICKET_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return CSSM_EHIP; }
  out[2 * fw + CSSM_PEER_TICKET_WORDS] = pf->peer_seq;
  return (int64_t)need;
}
// Rows the pack stages of this handle wrote for its neighbours since the windows were set up (per segment: the eager rows, or the needed
// ones where they were more), the number of neighbour segments they went into, and how many of those needed rows beyond the eager ones
// (where every row travels -- CSSM_PEER_ALL_ROWS=1 or an eager count >= the capacity -- that is the whole block per segment, none beyond)
extern "C" int cssm_pf_shard_peer_rows(cssm_pf* pf, uint64_t* rows_out, uint64_t* segments_out, uint64_t* beyond_out) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!rows_out || !segments_out) return fail(CSSM_EINVAL_ARG, "null argument");
  *rows_out = 0; *segments_out = 0;
  if (beyond_out) *beyond_out = 0;
  if (!pf->peer_tickets) return CSSM_OK;
  rc = bounded_sync(pf);
  if (rc) return rc;
  unsigned long long st[3] = {0ull, 0ull, 0ull};
  HIP_TRY(hipMemcpy(st, pf->peer_tickets + CSSM_PEER_TICKET_STAT, sizeof st, hipMemcpyDeviceToHost));
  *rows_out = st[0]; *segments_out = st[1];
  if (beyond_out) *beyond_out = st[2];
  return CSSM_OK;
}
extern "C" int cssm_pf_shard_adopt_peer(cssm_pf* pf, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->connected) return fail(CSSM_ESTATE, "peer windows are not set up (cssm_pf_shard_peer_setup / _connect)");
  if (rank != ps->rank || world != ps->world || cap != ps->cap) return fail(CSSM_ESHARD, "peer windows were set up for rank %d / world %d / cap %lld", ps->rank, ps->world, (long long)ps->cap);
  if (!pf->peer_packed) return fail(CSSM_ESTATE, "cssm_pf_shard_adopt_peer without cssm_pf_shard_pack_peer");
  if (!pf->peer_rows_packed) return fail(CSSM_ESTATE, "cssm_pf_shard_adopt_peer without cssm_pf_shard_pack_rows_peer (the rows are the pack's second stage)");
  pf->peer_packed = false; pf->peer_rows_packed = false;
  const int p = (int)(pf->peer_seq & 1u);
  return adopt_spec_impl(pf, peer_window(ps, ps->slab, p), rank, world, cap, peer_flagset(ps, ps->slab, p), pf->peer_seq);
}
// Pack and adopt in ONE launch (k_exchange_offspring): for a rank that has its stream to itself -- one process per GPU.  Several
// shards driven on one stream must use the two stage calls above, all packs before any adopt: a shard's pollers would wait for
// segments that a launch BEHIND them on the same stream is to write.
extern "C" int cssm_pf_shard_exchange_peer(cssm_pf* pf, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  PeerState* ps = static_cast<PeerState*>(pf->peer);
  if (!ps || !ps->connected) return fail(CSSM_ESTATE, "peer windows are not set up (cssm_pf_shard_peer_setup / _connect)");
  if (rank != ps->rank || world != ps->world || cap != ps->cap) return fail(CSSM_ESHARD, "peer windows were set up for rank %d / world %d / cap %lld", ps->rank, ps->world, (long long)ps->cap);
  if (pf->peer_packed) return fail(CSSM_ESTATE, "cssm_pf_shard_exchange_peer between cssm_pf_shard_pack_peer and cssm_pf_shard_adopt_peer");
  pf->peer_seq++;
  const int p = (int)(pf->peer_seq & 1u);
  return adopt_spec_impl(pf, peer_window(ps, ps->slab, p), rank, world, cap, peer_flagset(ps, ps->slab, p), pf->peer_seq, /*merged=*/true);
}

// The series loop on the peer-written exchange: per weighted observation propagate -> [pack into the peers' windows | offspring +
// expansion behind the flags] -- two launches, no collective, no host wait.  (Observations whose level comes from the global max --
// the first event of an LGCP series, a repeated series -- and resumed exchanges go through the RCCL / host-driven exchange.)
extern "C" int cssm_pf_shard_series_peer(cssm_pf* pf, int rank, int world, size_t s_begin, size_t s_end, const uint8_t* weighted, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!weighted) return fail(CSSM_EINVAL_ARG, "null argument");
  // no collective rides on this stretch: the waits behind it are plain stream waits (a communicator an EARLIER series of this handle
  // used -- bench.py's pre-flight runs one on every protocol -- would send them through bounded_sync's polling loop: 50 us sleeps and an
  // ncclCommGetAsyncError per poll, ~100 us per 20-observation leg)
  pf->last_comm = nullptr;
  for (size_t s = s_begin; s < s_end; ++s) {
    rc = cssm_pf_shard_propagate_at(pf, s, nullptr);
    if (rc) return rc;
    if (!weighted[s]) continue;
    static const bool two_launches = getenv("CSSM_PEER_TWO_LAUNCHES") != nullptr;   // (A/B: pack and adopt as launches of their own)
    if (two_launches) {
      rc = cssm_pf_shard_pack_peer(pf, rank, world, cap);
      if (rc) return rc;
      rc = cssm_pf_shard_pack_rows_peer(pf, rank, world, cap);
      if (rc) return rc;
      rc = cssm_pf_shard_adopt_peer(pf, rank, world, cap);
    } else {
      rc = cssm_pf_shard_exchange_peer(pf, rank, world, cap);
    }
    if (rc) return rc;
  }
  return CSSM_OK;
}

// ------------------------------------------------------------------------------------ series loop over RCCL, in the library
//
// The collectives of a single-collective series are driven from here instead of from the host language: per weighted
// observation  k_propagate<SUMS> -> k_boundary_pack -> ONE all-to-all -> k_offspring_expand_spec, all enqueued on the
// handle's stream without a host wait (through torch.distributed the same sequence costs several host-language calls per
// observation, which at 2^20 particles per GPU is longer than the kernels).  RCCL is resolved at run time -- the copy
// already loaded in the process (e.g. torch's) or librccl.so from the ROCm installation -- so the library itself links
// nothing but the HIP runtime.  Every wait for a stretch of the series is bounded (bounded_sync): a rank that never
// joins a collective, or an asynchronous RCCL error, aborts the communicator and surfaces as CSSM_ERCCL on this rank
// instead of hanging it.
#include <dlfcn.h>
namespace {
struct RcclApi {
  void* lib = nullptr;
  std::string path;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, cssm_rccl_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*AllToAll)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*AllToAllv)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int, void*, hipStream_t) = nullptr;
  int (*CommAbort)(void*) = nullptr;                 // optional
  int (*CommGetAsyncError)(void*, int*) = nullptr;   // optional
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
static RcclApi* rccl_api_load();
RcclApi* rccl_api() {   // (distinct handles may be driven from different threads: the first calls must not race on dlopen)
  static std::once_flag once;
  static RcclApi* loaded = nullptr;
  std::call_once(once, [] { loaded = rccl_api_load(); });
  return loaded;
}
static RcclApi* rccl_api_load() {
  static RcclApi api;
  // CSSM_RCCL_LIB: the library to take the nccl* entry points from, by path -- a particular RCCL build, or (tests/test_gpu_rccl_loopback.py)
  // a stand-in whose ranks are threads of one process, which drives this file's series loop at world > 1 on a single GPU
  if (const char* e = getenv("CSSM_RCCL_LIB")) {
    if (e[0]) { api.lib = dlopen(e, RTLD_NOW | RTLD_LOCAL); if (api.lib) api.path = e; else return nullptr; }
  }
  // the copy already mapped into this process (the host's framework usually brings one: two RCCL instances side by side
  // would each keep their own topology and IPC state), else the ROCm installation's
  if (FILE* maps = api.lib ? nullptr : fopen("/proc/self/maps", "r")) {
    char line[4096];
    while (!api.lib && fgets(line, sizeof line, maps)) {
      char* path = strchr(line, '/');
      if (!path || !strstr(path, "librccl.so")) continue;
      path[strcspn(path, "\n")] = 0;
      api.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (api.lib) api.path = path;
    }
    fclose(maps);
  }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* nm : names) {
    if (api.lib) break;
    api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) api.path = nm;
  }
  if (!api.lib) return nullptr;
  api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (int (*)(void**, int, cssm_rccl_id, int))dlsym(api.lib, "ncclCommInitRank");
  api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
  api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
  api.AllToAll = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllToAll");
  api.AllToAllv = (int (*)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int, void*, hipStream_t))dlsym(api.lib, "ncclAllToAllv");   // optional
  api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
  api.CommAbort = (int (*)(void*))dlsym(api.lib, "ncclCommAbort");
  api.CommGetAsyncError = (int (*)(void*, int*))dlsym(api.lib, "ncclCommGetAsyncError");
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllToAll;
  return api.ok ? &api : nullptr;
}
const int kNcclUint64 = 5, kNcclFloat64 = 8;   // ncclDataType_t (rccl.h)
int rccl_fail(RcclApi* a, const char* what, int r) {
  return fail(CSSM_ERCCL, "%s: %s", what, (a && a->GetErrorString) ? a->GetErrorString(r) : "RCCL error");
}
}  // namespace

// Wait for the handle's stream, but not forever once the library has put RCCL collectives on it: the wait polls, watches the
// communicator for asynchronous errors, and after CSSM_SHARD_TIMEOUT_S seconds (default 600) aborts the communicator --
// which ends the pending collectives on this rank -- and reports CSSM_ERCCL.  (Another rank died, or returned with an error
// before joining a collective: without this every surviving rank would sit in that collective for ever.)
static int bounded_sync(cssm_pf* pf) {
  if (!pf->last_comm) { HIP_TRY(hipStreamSynchronize(pf->stream)); return CSSM_OK; }
  RcclApi* a = rccl_api();
  double limit = 600.0;
  if (const char* e = getenv("CSSM_SHARD_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) limit = v; }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(pf->stream);
    if (e == hipSuccess) return CSSM_OK;
    if (e != hipErrorNotReady) return fail(CSSM_EHIP, "hipStreamQuery: %s", hipGetErrorString(e));
    int async_err = 0;
    const bool failed = a && a->CommGetAsyncError && a->CommGetAsyncError(pf->last_comm, &async_err) == 0 && async_err != 0;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed || waited > limit) {
      if (a && a->CommAbort) (void)a->CommAbort(pf->last_comm);
      pf->last_comm = nullptr;
      (void)hipStreamSynchronize(pf->stream);
      return failed ? rccl_fail(a, "asynchronous RCCL error in the sharded series", async_err)
                    : fail(CSSM_ERCCL, "a collective of the sharded series did not complete within %.0f s (a rank that never joined it?); "
                                       "the communicator was aborted", limit);
    }
    if (spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (the first polls spin: a stretch takes a few ms)
  }
}

extern "C" int cssm_rccl_available(void) { return rccl_api() ? 1 : 0; }
extern "C" const char* cssm_rccl_library(void) { RcclApi* a = rccl_api(); return a ? a->path.c_str() : ""; }

extern "C" int cssm_rccl_unique_id(cssm_rccl_id* id_out) {
  if (!id_out) return fail(CSSM_EINVAL_ARG, "null argument");
  RcclApi* a = rccl_api();
  if (!a) return fail(CSSM_ERCCL, "librccl.so could not be loaded");
  const int r = a->GetUniqueId(id_out);
  return r ? rccl_fail(a, "ncclGetUniqueId", r) : CSSM_OK;
}

extern "C" int cssm_rccl_comm_create(const cssm_rccl_id* id, int world, int rank, int device, void** comm_out) {
  if (!id || !comm_out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d", rank, world);
  RcclApi* a = rccl_api();
  if (!a) return fail(CSSM_ERCCL, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(device));
  void* comm = nullptr;
  const int r = a->CommInitRank(&comm, world, *id, rank);
  if (r) return rccl_fail(a, "ncclCommInitRank", r);
  *comm_out = comm;
  return CSSM_OK;
}

extern "C" void cssm_rccl_comm_destroy(void* comm) {
  RcclApi* a = rccl_api();
  if (a && comm) (void)a->CommDestroy(comm);
}

extern "C" int cssm_pf_shard_series_rccl(cssm_pf* pf, void* comm, int rank, int world, size_t s_begin, size_t s_end,
                                         const uint8_t* weighted, int64_t cap, uint64_t* sums5_dev, uint64_t* all_sums5_dev,
                                         double* send_buf_dev, double* recv_buf_dev, int single_collective) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!comm || !weighted || !sums5_dev || !all_sums5_dev || !send_buf_dev || !recv_buf_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  RcclApi* a = rccl_api();
  if (!a) return fail(CSSM_ERCCL, "librccl.so could not be loaded");
  pf->last_comm = comm;
  // single_collective & 4: the level comes from the GLOBAL max (LGCP; the repetition of a series an outlying observation
  // voided): an all-gather of the ranks' 5 words (only the max key matters) and cssm_pf_shard_sums precede the all-to-all
  const bool level_from_max = (single_collective & 4) != 0;
  single_collective &= 3;
  if (level_from_max && single_collective == 0) single_collective = 1;
  if (level_from_max && !a->AllGather) return fail(CSSM_ERCCL, "this RCCL has no ncclAllGather");
  if (single_collective) {   // sums and boundary particles in ONE all-to-all per observation (k_boundary_pack / k_expand_spec)
    const size_t sseg = (size_t)spec_seg(pf->d, (long long)cap);
    // single_collective == 2: only the two adjacent ranks get (and send) whole segments, every other pair exchanges the
    // 12-word segment header alone -- k_offspring_expand_spec reads nothing else of them (its verdict, formed from the
    // headers, rules out that a non-adjacent rank owns slots here).  Same call count, (world - 3) segments fewer on the
    // links per rank and observation.  Counts are a function of |rank - peer| only, so both ends of a pair agree.
    if (single_collective == 3 && !a->AllToAllv) return fail(CSSM_ERCCL, "this RCCL has no ncclAllToAllv");
    const bool trimmed = (single_collective == 3) || (single_collective == 2 && a->AllToAllv && world > 2);   // 3: tests (any world)
    std::vector<size_t> counts((size_t)world), displs((size_t)world);
    for (int q = 0; q < world; ++q) {
      counts[(size_t)q] = (q == rank + 1 || q == rank - 1) ? sseg : (size_t)kSpecHeaderWords;
      displs[(size_t)q] = (size_t)q * sseg;
    }
    // A rank that stops enqueueing -- an RCCL call failed, or one of its own stages did -- must end the collectives its peers
    // may already wait in: abort the communicator before returning (the peers' bounded waits then see an error, not a hang).
    auto bail = [&](int code) { if (a->CommAbort) (void)a->CommAbort(comm); pf->last_comm = nullptr; return code; };
    for (size_t s = s_begin; s < s_end; ++s) {
      rc = cssm_pf_shard_propagate_at(pf, s, level_from_max ? sums5_dev : nullptr);
      if (rc) return bail(rc);
      if (!weighted[s]) continue;
      if (level_from_max) {
        prof_begin(pf, CSSM_K_COLLECTIVE);
        const int rg = a->AllGather(sums5_dev, all_sums5_dev, 5, kNcclUint64, comm, pf->stream);
        prof_end(pf);
        if (rg) return bail(rccl_fail(a, "ncclAllGather", rg));
        rc = shard_sums_impl(pf, all_sums5_dev, world, nullptr);
        if (rc) return bail(rc);
      }
      rc = cssm_pf_shard_boundary_pack(pf, rank, world, cap, send_buf_dev);
      if (rc) return bail(rc);
      prof_begin(pf, CSSM_K_COLLECTIVE);
      const int r = trimmed ? a->AllToAllv(send_buf_dev, counts.data(), displs.data(), recv_buf_dev, counts.data(), displs.data(), kNcclFloat64,
                                           comm, pf->stream)
                            : a->AllToAll(send_buf_dev, recv_buf_dev, sseg, kNcclFloat64, comm, pf->stream);
      prof_end(pf);
      if (r) return bail(rccl_fail(a, trimmed ? "ncclAllToAllv" : "ncclAllToAll", r));
      rc = cssm_pf_shard_adopt_spec(pf, recv_buf_dev, rank, world, cap);
      if (rc) return bail(rc);
    }
    return CSSM_OK;
  }
  return fail(CSSM_EINVAL_ARG, "single_collective must be 1, 2 or 3 (+ 4: level from the all-gathered max)");
}

extern "C" int cssm_pf_shard_pack(cssm_pf* pf, int world, const int64_t* send_first_host, const int64_t* send_count_host,
                                  int skip_rank, double* send_buf_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!send_first_host || !send_count_host) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->send_first_dev) return fail(CSSM_ESTATE, "shard_pack before shard_offspring");
  int64_t total = 0;
  for (int q = 0; q < world; ++q) {   // the host copies are only checked; the kernel reads the device originals
    const int64_t c = send_count_host[q], f = send_first_host[q];
    if (c < 0 || f < 0 || (uint64_t)(f + c) > pf->n) return fail(CSSM_ESHARD, "send range [%lld, +%lld) outside the shard", (long long)f, (long long)c);
    if (q != skip_rank) total += c;
  }
  if (total > 0) {
    if (!send_buf_dev) return fail(CSSM_EINVAL_ARG, "send_buf_dev is null");
    hipLaunchKernelGGL(k_pack, dim3(grid_for((uint64_t)total, 256, kGridCap)), dim3(256), 0, pf->stream, pf->state[pf->cur], pf->stride,
                       pf->endslot, pf->d, world, pf->send_first_dev, pf->send_count_dev, (long long)total, skip_rank, send_buf_dev);
  }
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_adopt(cssm_pf* pf, const double* recv_buf_dev, int64_t n_low, int64_t n_high, int64_t self_first,
                                   int64_t self_count) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (n_low < 0 || n_high < 0 || self_count < 0 || self_first < 0 || (uint64_t)(self_first + self_count) > pf->n)
    return fail(CSSM_ESHARD, "bad candidate counts");
  const int64_t n_remote = n_low + n_high;
  if (n_remote + self_count < 1) return fail(CSSM_ESHARD, "a rank must have at least one candidate particle");
  if (n_remote > 0 && !recv_buf_dev) return fail(CSSM_EINVAL_ARG, "recv_buf_dev is null");
  if ((size_t)n_remote > pf->cand_cap) {
    HIP_TRY(hipStreamSynchronize(pf->stream));
    if (pf->cand) (void)hipFree(pf->cand);
    if (pf->cand_end) (void)hipFree(pf->cand_end);
    if (pf->cand_idx) (void)hipFree(pf->cand_idx);
    pf->cand = nullptr; pf->cand_end = pf->cand_idx = nullptr;
    size_t cap = (size_t)n_remote + (size_t)n_remote / 4 + CSSM_TILE;
    cap = (cap + CSSM_TILE - 1) / CSSM_TILE * CSSM_TILE;
    if (hipMalloc(&pf->cand, cap * 8 * pf->d + 64) != hipSuccess || hipMalloc(&pf->cand_end, cap * 4) != hipSuccess ||
        hipMalloc(&pf->cand_idx, cap * 4) != hipSuccess)
      return fail(CSSM_ENOMEM, "hipMalloc candidate buffers (%zu particles)", cap);
    pf->cand_cap = cap;
  }
  // The own particles' runs are already in anc (shard_offspring).  Candidates of lower ranks fill the slots below the
  // first own run (their last end slot is where it starts), candidates of higher ranks the slots from the last own
  // end slot upwards.
  const uint32_t n_split = (uint32_t)pf->n;
  if (n_remote > 0)
    hipLaunchKernelGGL(k_adopt_remote, dim3(grid_for((uint64_t)n_remote, 256, kGridCap)), dim3(256), 0, pf->stream, recv_buf_dev,
                       (long long)n_remote, pf->d, n_split, pf->cand, pf->cand_cap, pf->cand_end, pf->cand_idx);
  if (n_remote > 0)
    hipLaunchKernelGGL(k_expand, dim3(grid_for((uint64_t)n_remote, CSSM_BLOCK, kGridCap)), dim3(CSSM_BLOCK), 0, pf->stream, pf->cand_end,
                       pf->cand_idx, (uint64_t)n_remote, (uint64_t)n_low, pf->first, pf->first + pf->n, pf->anc,
                       (const uint32_t*)(pf->endslot + (pf->n - 1)));
  HIP_TRY(hipGetLastError());
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride;
  pf->src2 = pf->cand; pf->src2_stride = pf->cand_cap; pf->n_split = n_split; pf->anc_valid = true;
  if (pf->series) path_after(pf, last_rec_slot(pf));
  return CSSM_OK;
}

// ------------------------------------------------------------------------------------ cloud summaries over shards
// getIntervals (model/ParticleFilter.scala:415-424) of the SHARDED cloud: the order statistics are global ranks, so every
// radix-select pass totals the ranks' byte histograms with one all-reduce (the caller's) before every rank picks the same byte;
// the means are the all-reduced local sums over N_global.  Stage calls (ShardedFilter.summary drives them):
//   cssm_pf_shard_summary_begin   keys of the local resampled cloud (d state rows + eta = link(f(x, t))), local sums of the
//                                 state components -> sums_dev[d]; select states for the global ranks; hist_dev zeroed
//   8 x { cssm_pf_shard_summary_hist(shift) -> [all-reduce SUM of hist_dev: (d + 1) * 512 u32] -> cssm_pf_shard_summary_pick(shift) }
//   [all-reduce SUM of sums_dev]  cssm_pf_shard_summary_finish   means, order statistics, eta of the mean -> host
__global__ void k_partial_sums(const double* __restrict__ partial, int nblocks, int d, double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= d) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * d + k];
  out[k] = s;
}
extern "C" int cssm_pf_shard_summary_begin(cssm_pf* pf, double interval, double* sums_dev, uint32_t* hist_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!sums_dev || !hist_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  if (!pf->initialised) return fail(CSSM_ESTATE, "not initialised");
  if (!(interval > 0.0 && interval <= 1.0)) return fail(CSSM_EINVAL_ARG, "interval must be in (0, 1]");
  const int d = pf->d, rows = d + 1;
  const uint64_t n = pf->n, ng = pf->n_global;
  const int nblocks = grid_for(n, CSSM_BLOCK, 1024);
  if (pf->sm_cap < (size_t)n) {
    HIP_TRY(hipStreamSynchronize(pf->stream));
    void* old[] = {pf->sm_keys, pf->sm_partial, pf->sm_st, pf->sm_rec};
    for (void* q : old) if (q) (void)hipFree(q);
    pf->sm_keys = nullptr; pf->sm_partial = nullptr; pf->sm_st = nullptr; pf->sm_rec = nullptr; pf->sm_cap = 0;
    HIP_TRY(hipMalloc(&pf->sm_keys, (size_t)rows * n * 8));
    HIP_TRY(hipMalloc(&pf->sm_partial, (size_t)1024 * d * 8));
    HIP_TRY(hipMalloc(&pf->sm_st, rows * sizeof(SelState)));
    HIP_TRY(hipMalloc(&pf->sm_rec, sizeof(StepRec)));
    pf->sm_cap = (size_t)n;
  }
  pf->sm_blocks = nblocks; pf->sm_time = pf->t;
  // ranks, 0-based in ascending order, of the GLOBAL cloud: getCredibleInterval (:488-502) uses (N - index - 1, index - 1) with
  // index = floor(interval * N); getOrderStatistic (:455-460) uses (N - index, index)
  std::vector<SelState> hst(rows);
  const long long idxr = (long long)std::floor(interval * (double)ng);
  auto clampr = [&](long long r) { return (unsigned long long)std::min<long long>(std::max<long long>(r, 0), (long long)ng - 1); };
  for (int k = 0; k < rows; ++k) {
    hst[k].prefix[0] = hst[k].prefix[1] = 0;
    hst[k].rank[0] = clampr(k < d ? (long long)ng - idxr - 1 : (long long)ng - idxr);
    hst[k].rank[1] = clampr(k < d ? idxr - 1 : idxr);
  }
  StepRec hrec;
  cssm_build_rec(pf, pf->t, pf->t, 0.0, 0, pf->step, &hrec);   // F(t) of the cloud's time for f(x, t)
  HIP_TRY(hipMemcpyAsync(pf->sm_st, hst.data(), rows * sizeof(SelState), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(pf->sm_rec, &hrec, sizeof hrec, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));   // (the sources are stack objects)
  HIP_TRY(hipMemsetAsync(hist_dev, 0, (size_t)rows * 512 * 4, pf->stream));
  const uint32_t* idx = pf->anc_valid ? pf->anc : nullptr;
  DISPATCH_D(d, k_summary_fill<D><<<dim3(nblocks), dim3(CSSM_BLOCK), 0, pf->stream>>>(
                    pf->src, pf->src_stride, idx, idx ? pf->src2 : nullptr, pf->src2_stride, pf->n_split, n, (const StepRec*)pf->sm_rec, pf->mk,
                    pf->sm_keys, (size_t)n, pf->sm_partial));
  hipLaunchKernelGGL(k_partial_sums, dim3(1), dim3(64), 0, pf->stream, (const double*)pf->sm_partial, nblocks, d, sums_dev);
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}
extern "C" int cssm_pf_shard_summary_hist(cssm_pf* pf, int shift, uint32_t* hist_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!hist_dev || !pf->sm_keys || shift < 0 || shift > 56 || (shift & 7)) return fail(CSSM_EINVAL_ARG, "summary_hist: begin first; shift in {56, 48, ..., 0}");
  hipLaunchKernelGGL(k_sel_hist, dim3(pf->sm_blocks, pf->d + 1), dim3(CSSM_BLOCK), 0, pf->stream, (const unsigned long long*)pf->sm_keys, (size_t)pf->n, pf->n,
                     (const SelState*)pf->sm_st, shift, hist_dev);
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}
extern "C" int cssm_pf_shard_summary_pick(cssm_pf* pf, int shift, uint32_t* hist_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!hist_dev || !pf->sm_st) return fail(CSSM_EINVAL_ARG, "summary_pick: begin first");
  hipLaunchKernelGGL(k_sel_pick, dim3(pf->d + 1), dim3(2), 0, pf->stream, (SelState*)pf->sm_st, shift, hist_dev);   // (zeroes the histogram for the next pass)
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}
extern "C" int cssm_pf_shard_summary_finish(cssm_pf* pf, const double* sums_global_dev, double* state_mean, double* state_lower, double* state_upper,
                                            double* eta_of_mean, double* eta_lower, double* eta_upper) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!sums_global_dev || !pf->sm_st) return fail(CSSM_EINVAL_ARG, "summary_finish: begin first");
  const int d = pf->d, rows = d + 1;
  std::vector<SelState> hst(rows);
  std::vector<double> sums(d), mean(d);
  rc = bounded_sync(pf);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(hst.data(), pf->sm_st, rows * sizeof(SelState), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sums.data(), sums_global_dev, d * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < d; ++k) {
    mean[k] = sums[k] / (double)pf->n_global;
    if (state_mean) state_mean[k] = mean[k];
    if (state_lower) state_lower[k] = cssm_order_unkey(hst[k].prefix[0]);
    if (state_upper) state_upper[k] = cssm_order_unkey(hst[k].prefix[1]);
  }
  if (eta_lower) *eta_lower = cssm_order_unkey(hst[d].prefix[0]);
  if (eta_upper) *eta_upper = cssm_order_unkey(hst[d].prefix[1]);
  if (eta_of_mean) {
    StepRec hrec;
    cssm_build_rec(pf, pf->sm_time, pf->sm_time, 0.0, 0, pf->step, &hrec);
    *eta_of_mean = cssm_eta_of_mean(pf, hrec, mean.data());
  }
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_result(cssm_pf* pf, double* ll_out, int32_t* ess_out) {
  int rc = shard_check(pf);
  if (rc) return rc;
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (ll_out) *ll_out = h.ll;
  if (ess_out) *ess_out = h.ess;
  return cssm_check_device_err(pf, h);
}

#ifdef CSSM_OFF_STAMPS
// diagnostic build: the stamps k_offspring_expand_spec's blocks left (8 words per block)
extern "C" int cssm_pf_debug_spec_stamps(cssm_pf* pf, unsigned long long* out, size_t nwords) {
  if (!pf || nwords > 2048 * 8) return CSSM_EINVAL_ARG;
  HIP_TRY(hipStreamSynchronize(pf->stream));
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spec_stamps), nwords * 8));
  return CSSM_OK;
}
#endif

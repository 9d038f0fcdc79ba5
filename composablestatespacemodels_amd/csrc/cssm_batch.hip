// cssm_batch.hip -- B independent filters of ONE model structure advanced in lockstep: one launch per stage for all of them
// (grid.y = the chain).  The two chains of examples/DetermineParameters.scala:68-69 and the mapAsyncUnordered(4) pilot grid of
// model/Streaming.scala:38-39 run N = 100 000 particles each: a single chain's launches leave three quarters of the GPU idle and a
// step is launch latency + one wave's dependent instruction stream (LABNOTES_rounds1-3.md, section 5c); four host threads on four streams reached 510
// iterations/s against 178 for one chain.  Here ONE host thread enqueues two launches per observation for all B chains.
//
// What a chain keeps to itself: its buffers, its records (the parameters differ), its Philox key -- one ChainBase each, in a
// device array rewritten per call.  What the chains share: sizes, launch geometry, the record's index, buffer parities (every
// batched call starts from a fresh cloud, so the bookkeeping of all chains is the same).  A chain whose observation is ruled
// out of its reference level (err bit 6: its own later kernels return at once) or that the fused path does not serve is run
// again on its own through the single-handle driver: per-chain results are those of B single-handle runs, bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cssm_internal.h"
#include "cssm_kernels.hip.h"

struct cssm_pfb {
  std::vector<cssm_pf*> ch;
  int B = 0, device = 0;
  hipStream_t stream = nullptr;
  ChainBase* h_chains = nullptr;   // pinned
  ChainBase* d_chains = nullptr;
};

// ------------------------------------------------------------------------------------ kernels (grid.y = chain)

// ll = 0, ess = N, no error, no level predicted, every set of max slots and group sums clear: reset_scalars for every chain
static __global__ __launch_bounds__(CSSM_BLOCK) void k_reset_batch(const ChainBase* __restrict__ chains, int32_t ess0) {
  Scalars* sc = chains[blockIdx.y].sc;
  uint32_t* w = reinterpret_cast<uint32_t*>(sc);
  for (uint32_t i = blockIdx.x * CSSM_BLOCK + threadIdx.x; i < (uint32_t)(sizeof(Scalars) / 4); i += gridDim.x * CSSM_BLOCK) w[i] = 0u;
  (void)ess0;   // (the non-zero fields: k_reset_tail_batch, the next launch on the stream)
}
static __global__ void k_reset_tail_batch(const ChainBase* __restrict__ chains, int32_t ess0) {
  Scalars* sc = chains[blockIdx.x].sc;
  sc->ess = ess0; sc->fail_step = 0xffffffffu; sc->next_ref = cssm_nan();
}

template <int D>
__global__ __launch_bounds__(CSSM_BLOCK) void k_init_batch(const ChainBase* __restrict__ chains, size_t stride, uint64_t n, const double* __restrict__ logtab) {
  const ChainBase* __restrict__ c = chains + blockIdx.y;
  const double* tab = stage_log_table(logtab);
  double* dst = c->state[0];
  for (uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * CSSM_BLOCK) {
    double z[D];
    draw_normals<D>(c->seed, i, 0u, CSSM_STREAM_INIT, tab, z);
#pragma unroll
    for (int k = 0; k < D; ++k) dst[(size_t)k * stride + i] = c->sd0[k] * z[k] + c->m0[k];
  }
}

template <int RS, int RAWC, bool GRP>
__global__ __attribute__((amdgpu_flat_work_group_size(CSSM_BLOCK, CSSM_BLOCK), amdgpu_waves_per_eu(CSSM_OFF_WAVES, 8))) void k_offspring_batch(
    const ChainBase* __restrict__ chains, uint64_t n, uint32_t rec_idx, uint32_t ntiles, uint32_t sup, uint32_t nunits, int slot_set, int force_exact,
    int split, uint32_t s2_stride, int s2_par) {
  const ChainBase* __restrict__ c = chains + blockIdx.y;
  offspring_body<true, true, RS, RAWC, GRP>(c->logw, n, c->sc, c->tileS, c->tileS2, c->recs + rec_idx, n, nullptr, c->anc, ntiles, sup, nunits, RAWC, slot_set,
                                            c->ll_t, c->ess_t, rec_idx, force_exact, nullptr, 0, 1, split, c->seed, nullptr, nullptr, RAWC == 2 ? 1 : 0, nullptr, 0u,
                                            (uint32_t)n, 5u, c->s2buf, s2_stride, s2_par, c->gen);
}

static __global__ void k_record_batch(const ChainBase* __restrict__ chains, uint32_t s) {
  const ChainBase* c = chains + blockIdx.x;
  c->ll_t[s] = c->sc->ll; c->ess_t[s] = c->sc->ess;
}

// Resampling.sampleOne for `filter`, one particle of every chain's current cloud: row `row` of its path <- slot recs[pick_rec].pick
// (pick_rec < 0: the initial cloud's pick, the same function of the seed as the single-handle driver's)
static __global__ void k_pick_batch(const ChainBase* __restrict__ chains, int cur, int anc_valid, size_t stride, uint64_t n, int d, int pick_rec, uint32_t row) {
  const ChainBase* c = chains + blockIdx.x;
  const int k = threadIdx.x;
  if (k >= d) return;
  uint64_t idx;
  if (pick_rec < 0) {
    const int32_t pr = (int32_t)cssm_philox_draw(c->seed, 0, 0, CSSM_STREAM_PICK, 0).v[0];
    const uint32_t pa = pr < 0 ? (uint32_t)0 - (uint32_t)pr : (uint32_t)pr;
    idx = (uint64_t)pa % n;
  } else {
    idx = c->recs[pick_rec].pick;
  }
  const size_t j = anc_valid ? (size_t)c->anc[idx] : (size_t)idx;
  c->path[(size_t)row * d + k] = c->state[cur][(size_t)k * stride + j];
}

// k_finish for every chain: a pending ESS formed and filed, the scalars and the call's ll_t / ess_t into the chain's host-mapped
// mirrors, its completion word last
static __global__ __launch_bounds__(CSSM_BLOCK) void k_finish_batch(const ChainBase* __restrict__ chains, uint32_t s2_stride, uint32_t T, int want_t) {
  const ChainBase* __restrict__ c = chains + blockIdx.x;
  Scalars* sc = c->sc;
  __shared__ cssm_u128 s_red[CSSM_BLOCK / 64];
  __shared__ int32_t s_ess;
  const uint32_t pend = sc->pend, p_buf = sc->pend_buf, p_n = sc->pend_n, p_idx = sc->pend_idx, p_gen = sc->pend_gen, err = sc->err;
  const cssm_u128 p_S = sc->pend_S;
  const uint32_t gen = c->gen;
  int32_t ess = sc->ess;
  if (pend && p_buf < 2u && p_n <= s2_stride) {           // (uniform)
    cssm_u128 t2 = cssm_u128_zero();
    const cssm_u128* pb = c->s2buf + (size_t)p_buf * s2_stride;
    for (uint32_t q = threadIdx.x; q < p_n; q += CSSM_BLOCK) t2 = cssm_u128_add(t2, pb[q]);
    t2 = block_sum_u128(t2, s_red);
    if (threadIdx.x == 0) {
      if (!(err & 3u) && !cssm_u128_is_zero(p_S)) ess = cssm_ess_of(p_S, t2);
      sc->ess = ess; sc->pend = 0u;
      if (p_gen == gen && p_idx < T) c->ess_t[p_idx] = ess;
      s_ess = ess;
    }
    __syncthreads();
    ess = s_ess;
  }
  if (want_t) {
    for (uint32_t s = threadIdx.x; s < T; s += CSSM_BLOCK) c->host_ll_t[s] = c->ll_t[s];
    for (uint32_t s = threadIdx.x; s < T; s += CSSM_BLOCK) c->host_ess_t[s] = (pend && p_gen == gen && s == p_idx) ? ess : c->ess_t[s];
  }
  constexpr uint32_t W0 = (uint32_t)(offsetof(Scalars, err) / 4), W1 = (uint32_t)(sizeof(Scalars) / 4);
  constexpr uint32_t WE = (uint32_t)(offsetof(Scalars, ess) / 4), WP = (uint32_t)(offsetof(Scalars, pend) / 4);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(sc);
  uint32_t* dst = reinterpret_cast<uint32_t*>(c->host_sc);
  for (uint32_t w = W0 + threadIdx.x; w < W1; w += CSSM_BLOCK) dst[w] = (w == WE) ? (uint32_t)ess : ((w == WP) ? 0u : src[w]);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(c->host_done, c->done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------ host

// One observation of all chains: k_propagate_batch (through the dispatcher of the chain's latent dimension: cssm_launch_propagate with
// the chain table) and, if it is weighted, k_offspring_batch.  The bookkeeping is chain 0's, kept by the same code as a single handle's.
static int batch_step(cssm_pf* p0, const ChainBase* d_chains, int B, uint32_t s, int weighted, bool want_path) {
  p0->h_step_for_resample = s;
  CssmBatchLaunch bl;
  bl.chains = d_chains; bl.nchains = B; bl.rec_idx = s; bl.want_pick = want_path ? 1 : 0;
  int rc = cssm_launch_propagate(p0, p0->d_recs + s, nullptr, 0, &bl);
  if (rc || !weighted) return rc;
  if (!p0->last_optimistic) return fail(CSSM_ESTATE, "a batched launch must be a fused-sums launch");
  const int grid = (int)p0->nunits + 1;   // one block per unit + the publisher
  const int s2_par = p0->s2_par;
#define OFFB_ARGS d_chains, p0->n, s, p0->ntiles, p0->sup, p0->nunits, p0->wparity, p0->opt_exact, (int)p0->split, p0->s2_stride, s2_par
  if (p0->last_grp) hipLaunchKernelGGL((k_offspring_batch<CSSM_RESAMPLE_SYSTEMATIC, 2, true>), dim3(grid, B), dim3(CSSM_BLOCK), 0, p0->stream, OFFB_ARGS);
  else hipLaunchKernelGGL((k_offspring_batch<CSSM_RESAMPLE_SYSTEMATIC, 2, false>), dim3(grid, B), dim3(CSSM_BLOCK), 0, p0->stream, OFFB_ARGS);
#undef OFFB_ARGS
  HIP_TRY(hipGetLastError());
  p0->wparity = (p0->wparity + 1) % CSSM_MAXSETS;
  p0->anc_valid = true; p0->wmode = true; p0->have_level = true;
  p0->s2_par ^= 1;
  return CSSM_OK;
}
// the other chains went through exactly what chain 0's bookkeeping records
static void batch_copy_state(cssm_pf* pf, const cssm_pf* p0) {
  pf->cur = p0->cur; pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->src2 = nullptr; pf->anc_valid = p0->anc_valid;
  pf->wparity = p0->wparity; pf->wmode = p0->wmode; pf->s2_par = p0->s2_par; pf->last_optimistic = p0->last_optimistic; pf->last_grp = p0->last_grp;
  pf->have_level = p0->have_level; pf->sums_ready = p0->sums_ready;
}

extern "C" void cssm_pfb_destroy(cssm_pfb* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (cssm_pf* pf : b->ch) cssm_pf_destroy(pf);
  if (b->h_chains) (void)hipHostFree(b->h_chains);
  if (b->d_chains) (void)hipFree(b->d_chains);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

extern "C" int cssm_pfb_create(const cssm_model_desc* desc, uint64_t n_particles, int n_chains, int device, cssm_pfb** out) {
  if (!out) return fail(CSSM_EINVAL_ARG, "out is null");
  *out = nullptr;
  if (n_chains < 1 || n_chains > 64) return fail(CSSM_EINVAL_ARG, "1 .. 64 chains per batch");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(CSSM_EINVAL_ARG, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  cssm_pfb* b = new cssm_pfb();
  b->B = n_chains; b->device = device;
  auto bail = [&](int rc) { const std::string keep = cssm_last_error(); cssm_pfb_destroy(b); return fail(rc, "%s", keep.c_str()); };
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(CSSM_EHIP, "hipStreamCreate"));
  for (int k = 0; k < n_chains; ++k) {
    cssm_pf* pf = nullptr;
    int rc = cssm_pf_create_on_stream(desc, n_particles, 0, device, b->stream, &pf);   // (the chains share the batch's stream)
    if (rc) return bail(rc);
    b->ch.push_back(pf);
  }
  if (hipHostMalloc((void**)&b->h_chains, sizeof(ChainBase) * n_chains, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(&b->d_chains, sizeof(ChainBase) * n_chains) != hipSuccess)
    return bail(fail(CSSM_ENOMEM, "chain table"));
  *out = b;
  return CSSM_OK;
}

extern "C" int cssm_pfb_num_chains(const cssm_pfb* b) { return b ? b->B : 0; }
extern "C" cssm_pf* cssm_pfb_chain(cssm_pfb* b, int k) { return (b && k >= 0 && k < b->B) ? b->ch[(size_t)k] : nullptr; }

// whether the batched launches serve this handle's configuration (else every chain runs through the single-handle driver)
static bool batch_serves(const cssm_pf* pf) { return cssm_batch_ok(pf) != 0; }

// `filter` (model/ParticleFilter.scala:152-158) of every chain: chain k under descs[k] (the structure the batch was created with)
// and seeds[k]; ll_out[k], path_out (may be NULL) [k][(T + 1) * d], rc_out[k] = the chain's own status (CSSM_ENONFINITE: its weights
// were unusable -- a PMMH proposal the filter cannot weigh).  Returns non-zero only for errors that are not a chain's own.
extern "C" int cssm_pfb_filter(cssm_pfb* b, const cssm_model_desc* const* descs, const uint64_t* seeds, const double* t, const double* y,
                               const uint8_t* has, size_t T, double* ll_out, double* path_out, int* rc_out) {
  if (!b || !descs || !seeds || !t || !y || !ll_out || !rc_out) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data (the reference's minBy throws on an empty Vector)");
  HIP_TRY(hipSetDevice(b->device));
  const int B = b->B;
  cssm_pf* p0 = b->ch[0];
  const int d = p0->d;
  const bool want_path = path_out != nullptr;
  for (int k = 0; k < B; ++k) {
    rc_out[k] = CSSM_OK;
    int rc = cssm_pf_set_params(b->ch[(size_t)k], descs[k]);
    if (rc) return rc;
    (void)cssm_pf_reseed(b->ch[(size_t)k], seeds[k]);
  }
  auto run_alone = [&](int k) {   // the chain through the single-handle driver (its own init, its own redo of a held observation)
    cssm_pf* pf = b->ch[(size_t)k];
    double ll = 0.0;
    const int rc = want_path ? cssm_pf_filter(pf, t, y, has, T, &ll, nullptr, nullptr, path_out + (size_t)k * (T + 1) * d)
                             : cssm_pf_ll_filter(pf, t, y, has, T, &ll, nullptr, nullptr);
    ll_out[k] = ll; rc_out[k] = rc;
  };
  if (!batch_serves(p0) || B == 1) {
    for (int k = 0; k < B; ++k) run_alone(k);
    return CSSM_OK;
  }
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];
  // records (the chains' parameters differ), the chain table
  for (int k = 0; k < B; ++k) {
    cssm_pf* pf = b->ch[(size_t)k];
    int rc = cssm_ensure_recs(pf, T);
    if (rc) return rc;
    if (want_path && pf->path_cap < (T + 1) * (size_t)d) {
      HIP_TRY(hipStreamSynchronize(b->stream));
      if (pf->d_path) (void)hipFree(pf->d_path);
      pf->d_path = nullptr;
      HIP_TRY(hipMalloc(&pf->d_path, (T + 1) * (size_t)d * 8));
      pf->path_cap = (T + 1) * (size_t)d;
    }
    double tp = t0;
    for (size_t s = 0; s < T; ++s) { cssm_build_rec(pf, tp, t[s], y[s], has ? has[s] : 1, (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
    rc = cssm_upload_recs(pf, 0, T, false);
    if (rc) return rc;
    cssm_batch_fresh(pf, t0);                 // host-side state of a freshly drawn cloud; gen, done_seq advanced
    ChainBase& c = b->h_chains[k];
    c.state[0] = pf->state[0]; c.state[1] = pf->state[1]; c.logw = pf->logw; c.anc = pf->anc; c.tileS = pf->tileS; c.tileS2 = pf->tileS2;
    c.sc = pf->sc; c.recs = pf->d_recs; c.s2buf = pf->s2buf; c.ll_t = pf->d_ll_t; c.ess_t = pf->d_ess_t; c.path = pf->d_path;
    c.m0 = pf->d_m0; c.sd0 = pf->d_sd0; c.host_sc = pf->hd_sc; c.host_ll_t = pf->hd_ll_t; c.host_ess_t = pf->hd_ess_t; c.host_done = pf->hd_done;
    c.seed = pf->seed; c.gen = pf->gen; c.done_seq = pf->done_seq;
  }
  HIP_TRY(hipMemcpyAsync(b->d_chains, b->h_chains, sizeof(ChainBase) * B, hipMemcpyHostToDevice, b->stream));
  const uint64_t n = p0->n;
  const int32_t ess0 = (int32_t)(n < 2147483647ull ? n : 2147483647ull);
  hipLaunchKernelGGL(k_reset_batch, dim3(8, B), dim3(CSSM_BLOCK), 0, b->stream, (const ChainBase*)b->d_chains, ess0);
  hipLaunchKernelGGL(k_reset_tail_batch, dim3(B), dim3(1), 0, b->stream, (const ChainBase*)b->d_chains, ess0);
  DISPATCH_D(d, k_init_batch<D><<<dim3(grid_for(n, CSSM_BLOCK, 1024), B), dim3(CSSM_BLOCK), 0, b->stream>>>((const ChainBase*)b->d_chains, p0->stride, n, p0->d_logtab));
  if (want_path) hipLaunchKernelGGL(k_pick_batch, dim3(B), dim3(64), 0, b->stream, (const ChainBase*)b->d_chains, 0, 0, p0->stride, n, d, -1, 0u);
  HIP_TRY(hipGetLastError());
  // the observations: two launches for all chains each.  The bookkeeping (buffer parity, max-slot set, ESS buffer) is chain 0's,
  // advanced by the same helpers as the single-handle driver; it is copied to the other chains at the end.
  for (size_t s = 0; s < T; ++s) {
    const int weighted = p0->h_recs[s].has_obs;
    int rc = batch_step(p0, b->d_chains, B, (uint32_t)s, weighted, want_path);
    if (rc) return rc;
    if (!weighted) hipLaunchKernelGGL(k_record_batch, dim3(B), dim3(1), 0, b->stream, (const ChainBase*)b->d_chains, (uint32_t)s);
  }
  if (want_path)   // (the last entry has no following propagate to record it)
    hipLaunchKernelGGL(k_pick_batch, dim3(B), dim3(64), 0, b->stream, (const ChainBase*)b->d_chains, p0->cur, p0->anc_valid ? 1 : 0, p0->stride, n, d,
                       (int)(T - 1), (uint32_t)T);
  if (want_path)
    for (int k = 0; k < B; ++k)
      HIP_TRY(hipMemcpyAsync(path_out + (size_t)k * (T + 1) * d, b->ch[(size_t)k]->d_path, (T + 1) * (size_t)d * 8, hipMemcpyDeviceToHost, b->stream));
  hipLaunchKernelGGL(k_finish_batch, dim3(B), dim3(CSSM_BLOCK), 0, b->stream, (const ChainBase*)b->d_chains, p0->s2_stride, (uint32_t)T, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int k = 0; k < B; ++k) {
    cssm_pf* pf = b->ch[(size_t)k];
    if (k) batch_copy_state(pf, p0);
    pf->t = t[T - 1]; pf->step = (uint32_t)T;
    const Scalars& h = *pf->h_sc;
    pf->ess_host = h.ess;
    if ((h.err & 64u) && !(h.err & 3u)) { run_alone(k); continue; }   // an observation was ruled out of its level: this chain again, on its own
    ll_out[k] = h.ll;
    rc_out[k] = cssm_check_device_err(pf, h);
  }
  return CSSM_OK;
}

// ParticleMetropolisHastings (model/PMMH.scala:68-81,114-123) for B chains in lockstep: chain k is cssm_pmmh_run with theta0[k] and
// seeds[k] -- the same proposals, the same filter keys, the same accept / reject decisions, bit for bit -- but every iteration's B
// filters run as ONE batch.  Outputs chain-major: ll[k * n_iters + it], theta[(k * n_iters + it) * n_theta + j], ...
extern "C" int cssm_pmmh_run_batched(cssm_pfb* b, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, const double* t,
                                     const double* y, const uint8_t* has, size_t T, const uint64_t* seeds, size_t n_iters, double* ll, double* theta,
                                     int32_t* accepted, double* last_state) {
  if (!b || !desc || !theta0 || !seeds || !ll || !theta || !accepted || !last_state) return fail(CSSM_EINVAL_ARG, "null argument");
  const int B = b->B, d = b->ch[0]->d;
  std::vector<cssm_pmmh_chain*> chains((size_t)B, nullptr);
  std::vector<const cssm_model_desc*> descs((size_t)B);
  std::vector<uint64_t> keys((size_t)B);
  std::vector<double> pll((size_t)B), paths((size_t)B * (T + 1) * d);
  std::vector<int> rcs((size_t)B);
  int rc = CSSM_OK;
  for (int k = 0; k < B && !rc; ++k) rc = cssm_pmmh_chain_create(desc, theta0 + (size_t)k * n_theta, n_theta, delta, seeds[k], d, &chains[(size_t)k]);
  for (size_t it = 0; it < n_iters && !rc; ++it) {
    for (int k = 0; k < B; ++k) descs[(size_t)k] = cssm_pmmh_chain_propose(chains[(size_t)k], it, &keys[(size_t)k]);
    rc = cssm_pfb_filter(b, descs.data(), keys.data(), t, y, has, T, pll.data(), paths.data(), rcs.data());
    if (rc) break;
    for (int k = 0; k < B; ++k) {
      if (rcs[(size_t)k] == CSSM_ENONFINITE) pll[(size_t)k] = -cssm_inf();     // a proposal the filter cannot weigh is rejected
      else if (rcs[(size_t)k]) { rc = rcs[(size_t)k]; break; }
      const size_t o = (size_t)k * n_iters + it;
      cssm_pmmh_chain_decide(chains[(size_t)k], it, pll[(size_t)k], paths.data() + ((size_t)k * (T + 1) + T) * d, &ll[o], theta + o * n_theta, &accepted[o],
                             last_state + o * (size_t)d);
    }
  }
  for (cssm_pmmh_chain* c : chains) cssm_pmmh_chain_destroy(c);
  return rc;
}

// ONE chain, two iterations per batch of three filters.  The chain is sequential -- iteration i + 1 proposes from whatever iteration i
// leaves -- but it leaves one of two things: its proposal (accepted) or the parameters it started from (rejected).  Proposals and filter
// keys are functions of (seed, iteration, parameters) alone (counter-based streams: cssm_pmmh_chain_propose), so BOTH candidates for
// iteration i + 1 can be drawn and filtered before iteration i has decided: slot 0 filters iteration i's proposal, slot 1 iteration i + 1's
// proposal from it, slot 2 iteration i + 1's proposal from the current parameters -- one batch, at N = 100 000 for little more than the
// price of one filter (a single filter leaves most of the GPU idle) -- and the two decisions follow, the second on slot 1 or 2.  Output
// identical to cssm_pmmh_run(seed), bit for bit, on its error paths too (a batch that fails is redone candidate by candidate, in the
// sequential chain's order); b holds exactly three chains.  model/PMMH.scala:68-81,114-123.
extern "C" int cssm_pmmh_run_speculative(cssm_pfb* b, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, const double* t,
                                         const double* y, const uint8_t* has, size_t T, uint64_t seed, size_t n_iters, double* ll, double* theta,
                                         int32_t* accepted, double* last_state) {
  if (!b || !desc || !theta0 || !ll || !theta || !accepted || !last_state) return fail(CSSM_EINVAL_ARG, "null argument");
  const int B = b->B, d = b->ch[0]->d;
  if (B != 3) return fail(CSSM_EINVAL_ARG, "the speculative chain filters exactly three candidates per batch: a batch of %d chains (further slots would re-filter a proposal for nothing)", B);
  // the chain itself (L) and one proposer per slot; all under the chain's seed -- the proposal of iteration `it` from parameters p is the
  // same numbers whoever draws it
  cssm_pmmh_chain* L = nullptr;
  std::vector<cssm_pmmh_chain*> slot((size_t)B, nullptr);
  std::vector<const cssm_model_desc*> descs((size_t)B);
  std::vector<uint64_t> keys((size_t)B);
  std::vector<double> pll((size_t)B), paths((size_t)B * (T + 1) * d);
  std::vector<int> rcs((size_t)B);
  int rc = cssm_pmmh_chain_create(desc, theta0, n_theta, delta, seed, d, &L);
  for (int k = 0; k < B && !rc; ++k) rc = cssm_pmmh_chain_create(desc, theta0, n_theta, delta, seed, d, &slot[(size_t)k]);
  auto used = [&](int k, double* pl) -> int {   // the slot's likelihood as the sequential driver would have seen it
    *pl = pll[(size_t)k];
    if (rcs[(size_t)k] == CSSM_ENONFINITE) { *pl = -cssm_inf(); return CSSM_OK; }   // a proposal the filter cannot weigh is rejected
    return rcs[(size_t)k];
  };
  for (size_t it = 0; it < n_iters && !rc; it += 2) {
    const bool two = it + 1 < n_iters;
    cssm_pmmh_chain_set_current(slot[0], cssm_pmmh_chain_current(L));
    descs[0] = cssm_pmmh_chain_propose(slot[0], it, &keys[0]);
    for (int k = 1; k < B; ++k) {
      // slot 1: iteration it + 1 from iteration it's proposal (it is accepted); slot 2: from the current parameters (it is rejected);
      // an odd last iteration and further slots: iteration it's proposal again
      const bool next = two && k <= 2;
      cssm_pmmh_chain_set_current(slot[(size_t)k], (next && k == 1) ? cssm_pmmh_chain_proposal(slot[0]) : cssm_pmmh_chain_current(L));
      descs[(size_t)k] = cssm_pmmh_chain_propose(slot[(size_t)k], next ? it + 1 : it, &keys[(size_t)k]);
    }
    // the candidate the sequential chain would filter next, alone (every slot the same proposal): the fall-back where the batch failed for a
    // reason that may belong to a candidate the chain never evaluates -- a speculative proposal whose parameters the library refuses, say
    auto alone = [&](int k) -> int {
      for (int j = 0; j < B; ++j) { descs[(size_t)j] = descs[(size_t)k]; keys[(size_t)j] = keys[(size_t)k]; }
      const int r = cssm_pfb_filter(b, descs.data(), keys.data(), t, y, has, T, pll.data(), paths.data(), rcs.data());
      if (r) return r;
      pll[(size_t)k] = pll[0]; rcs[(size_t)k] = rcs[0];
      if (k) memcpy(paths.data() + (size_t)k * (T + 1) * d, paths.data(), (T + 1) * (size_t)d * 8);
      return CSSM_OK;
    };
    const cssm_model_desc* d1 = descs[1]; const cssm_model_desc* d2 = descs[2];
    const uint64_t k1 = keys[1], k2 = keys[2];
    bool batch_ok = true;
    rc = cssm_pfb_filter(b, descs.data(), keys.data(), t, y, has, T, pll.data(), paths.data(), rcs.data());
    if (rc) { batch_ok = false; rc = alone(0); }               // (a failure of slot 0's own candidate fails again: the sequential chain's error)
    if (rc) break;
    double pl = 0.0;
    rc = used(0, &pl);
    if (rc) break;
    const int32_t before = cssm_pmmh_chain_accepted(L);
    cssm_pmmh_chain_set_proposal(L, cssm_pmmh_chain_proposal(slot[0]));
    cssm_pmmh_chain_decide(L, it, pl, paths.data() + ((size_t)0 * (T + 1) + T) * d, &ll[it], theta + it * n_theta, &accepted[it], last_state + it * (size_t)d);
    if (!two) break;
    const int k = (cssm_pmmh_chain_accepted(L) > before) ? 1 : 2;
    if (!batch_ok) {                                           // the batch never ran: the needed candidate of iteration it + 1 on its own
      descs[1] = d1; descs[2] = d2; keys[1] = k1; keys[2] = k2;
      rc = alone(k);
      if (rc) break;
    }
    rc = used(k, &pl);
    if (rc) break;
    cssm_pmmh_chain_set_proposal(L, cssm_pmmh_chain_proposal(slot[(size_t)k]));
    cssm_pmmh_chain_decide(L, it + 1, pl, paths.data() + ((size_t)k * (T + 1) + T) * d, &ll[it + 1], theta + (it + 1) * n_theta, &accepted[it + 1],
                           last_state + (it + 1) * (size_t)d);
  }
  cssm_pmmh_chain_destroy(L);
  for (cssm_pmmh_chain* c : slot) cssm_pmmh_chain_destroy(c);
  return rc;
}

// cssm_pf.hip -- host side of libcssm_pf: the C ABI of include/cssm_pf.h over the gfx950
// kernels of cssm_kernels.hip.h.  No torch, no CPU compute path: every entry point drives HIP.
#include <hip/hip_runtime.h>
#include <cstddef>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "cssm_host.h"
#include "cssm_kernels.hip.h"
#include "cssm_series_abi.h"

// ------------------------------------------------------------------------------------ errors

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) return fail(CSSM_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

extern "C" const char* cssm_last_error(void) { return g_err.c_str(); }
extern "C" const char* cssm_version(void) { return "cssm_pf 0.3 (gfx950, numerics contract v5)"; }

// ------------------------------------------------------------------------------------ handle

#define CSSM_NKERNELS CSSM_PROFILE_NKERNELS

struct Comp {
  int kind, leaf, idx, f_kind, period;
  double m0, c0, mu, phi, sigma;
};

struct cssm_pf {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  // model
  int d = 0, n_leaves = 0, obs_kind = 0, precision = 0, obs_df = 0;
  double scale_sd = 1.0;       // exp(scale): Gaussian sd, NegBin size, Student-t v
  double scale_raw = 0.0;      // ZIP: the stored scale v
  Comp comp[CSSM_MAX_DIM];
  ModelK mk;
  // sizes
  uint64_t n_global = 0, first = 0, n = 0, seed = 0;
  size_t stride = 0;
  uint32_t ntiles = 0;
  uint32_t sup = 1, nunits = 0;   // tiles per scan unit, number of units (<= ~1K)
  uint32_t split = 1;             // k_propagate blocks per unit (each owns a contiguous sub-unit and its sums)
  bool safe_sums = false;         // form the sums in their own pass after the max is known (retry of a step whose
                                  // reference level was ruled out by the max; always for LGCP)
  bool last_optimistic = false;   // the last launch_propagate formed the sums itself
  void* last_comm = nullptr;      // RCCL communicator the library last enqueued collectives on (bounded_sync)
  bool lgcp_tdep = false;         // LGCP whose f depends on time (a seasonal leaf): f is evaluated at every sub-step time (d_fsub)
  std::vector<double> h_fsub;     // host copy of the sub-step coefficient table of the records last built
  double* d_fsub = nullptr; size_t fsub_cap = 0;
  bool batch_hold = false;        // batch drivers: an outlying observation puts the series on hold (err bit 6) instead of voiding it
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
  int wparity = 0;             // max-slot set (0 .. CSSM_MAXSETS - 1) of the next weighted step (single-GPU path)
  int opt_exact = 0;           // CSSM_OPT_EXACT_OFFSPRING
  int opt_fused = 0;           // CSSM_OPT_FUSED_SUMS (set to 1 for sharded handles at creation)
  int opt_step = 0;            // CSSM_OPT_ONE_LAUNCH: 0 = never (default: measured no faster), 1 = whenever eligible, -1 = for clouds up to CSSM_STEP_MAX_N
  double* logw_alt = nullptr;  // k_step reads the log-weights / unit sums of observation s - 1 and writes those of s: two sets,
  cssm_u128 *tileS_alt = nullptr, *tileS2_alt = nullptr;   //   logw / tileS / tileS2 always being the set written last
  int pp = 0;                  // how often the sets were swapped since launch_init, mod 2
  cssm_u128 *fineS = nullptr, *fineS2 = nullptr;   // large clouds: the sums of k_propagate's single-tile blocks (k_reduce_units folds them into tileS / tileS2)
  size_t fine_cap = 0;
  bool no_fine = false;        // (k_step in use: no extra launch between the kernels it merges)
  int opt_whole = 0;           // CSSM_OPT_WHOLE_TILES
  std::vector<uint8_t> pp_after;   // pp right after the propagate of every observation of the batch run (restored when a series is put on hold)
  int opt_series = 0;          // CSSM_OPT_SERIES_KERNEL: 1 = batch drivers run the persistent series kernel when the handle is eligible (opt-in, see cssm_pf.h)
  // persistent series kernel (cssm_series.hip.h)
  void* d_sync = nullptr;      // SeriesSync
  int ser_blocks_max = -1;     // CUs of a device that can launch cooperatively (-1: not asked yet, 0: it cannot)
  std::vector<int> ser_occ;    // resident blocks per CU by particles per block / tile (-1: not asked yet)
  unsigned long long* d_ts = nullptr;   // profiling: block 0's timestamps, 5 per observation
  size_t ts_cap = 0;
  int profile_level = 0;       // 2: every block of the series kernel stamps its phases (cssm_pf_series_stamps)
  uint32_t ts_blocks = 0; size_t ts_T = 0;
  bool last_series = false;    // the last batch run used the series kernel
  double ser_phase_us[4] = {0, 0, 0, 0};   // profiling: average phase P / exchange / phase O / closing barrier of the weighted steps
  uint64_t ser_phase_steps = 0;
  int resampler = CSSM_RESAMPLE_SYSTEMATIC;
  double* cum = nullptr;       // multinomial: cumulative normalised weights
  const long long *send_first_dev = nullptr, *send_count_dev = nullptr;   // last shard_offspring outputs (device)
  cssm_u128 *tileS = nullptr, *tileS2 = nullptr, *tileP = nullptr;
  Scalars* sc = nullptr;
  double *d_m0 = nullptr, *d_sd0 = nullptr, *d_logtab = nullptr;
  StepRec* d_recs = nullptr;
  size_t recs_cap = 0;
  double* d_ll_t = nullptr;
  int32_t* d_ess_t = nullptr;
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
  bool series = false;         // records of a whole series are resident (cssm_pf_shard_begin)
  struct Snap { int cur; const double* src; size_t src_stride; const double* src2; size_t src2_stride; uint32_t n_split; bool anc_valid, last_optimistic; uint32_t step; double t; };
  std::vector<Snap> snaps;     // host-side state right after the propagate of every observation of the series (cssm_pf_shard_resume)
  // host staging (pinned)
  StepRec* h_recs = nullptr;
  size_t h_recs_cap = 0;
  Scalars* h_sc = nullptr;     // pinned: the streaming step's scalars land here without a staging copy
  // filter state
  double t = 0.0;
  uint32_t step = 0;
  uint32_t h_step_for_resample = 0;   // observation index of the step being resampled (Philox counter word)
  bool initialised = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  // optional per-kernel timing (HIP events on the launch stream around every kernel)
  bool profile = false;
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
static void prof_collect(cssm_pf* pf) {
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

// ------------------------------------------------------------------------------------ model

static double rep(const double* v, int n, int i) { return v[i % n]; }               // Sde.buildParamRepeat, model/Sde.scala:177-179
static double logistic(double x) { return 1.0 / (1.0 + cssm_exp(-x)); }              // model/SdeParameters.scala:214-216

// Constraint transforms of the SDE constructors: model/Sde.scala:70-73 (GenBrownian),
// :99-102 (Brownian), :133-137 (OU, logistic applied to the stored value).
static int build_model_into(cssm_pf* pf, const cssm_model_desc* desc);

// Validate and translate the descriptor into a scratch handle first; the real handle changes only when everything
// checked out (a failing cssm_pf_set_params leaves the previous parameters in force, whole).  `update`: the call
// re-parameterises an existing handle -- the STRUCTURE (leaves, dimensions, SDE kinds, f kinds, periods, observation
// model, LGCP precision, Student-t df) must be the one the handle was created with: kernels, buffers and the sharding of
// work were sized for it.
static int build_model(cssm_pf* pf, const cssm_model_desc* desc, bool update) {
  cssm_pf tmp;
  int rc = build_model_into(&tmp, desc);
  if (rc) return rc;
  if (update) {
    bool same = tmp.d == pf->d && tmp.n_leaves == pf->n_leaves && tmp.obs_kind == pf->obs_kind && tmp.precision == pf->precision &&
                tmp.obs_df == pf->obs_df;
    for (int k = 0; same && k < tmp.d; ++k) {
      const Comp &a = tmp.comp[k], &b = pf->comp[k];
      same = a.kind == b.kind && a.leaf == b.leaf && a.idx == b.idx && a.f_kind == b.f_kind && a.period == b.period;
    }
    if (!same) return fail(CSSM_EINVAL_DESC, "set_params: the model structure differs from the one the handle was created with "
                                             "(only parameter values may change; create a new handle for another model)");
  }
  pf->d = tmp.d; pf->n_leaves = tmp.n_leaves; pf->obs_kind = tmp.obs_kind; pf->precision = tmp.precision; pf->obs_df = tmp.obs_df;
  pf->scale_sd = tmp.scale_sd; pf->scale_raw = tmp.scale_raw; pf->mk = tmp.mk;
  pf->lgcp_tdep = false;
  if (tmp.obs_kind == CSSM_OBS_LGCP) for (int k = 0; k < tmp.d; ++k) if (tmp.comp[k].f_kind == CSSM_F_SEASONAL) pf->lgcp_tdep = true;
  for (int k = 0; k < tmp.d; ++k) pf->comp[k] = tmp.comp[k];
  return CSSM_OK;
}

static int build_model_into(cssm_pf* pf, const cssm_model_desc* desc) {
  if (!desc || !desc->leaves) return fail(CSSM_EINVAL_DESC, "null model descriptor");
  if (desc->n_leaves < 1 || desc->n_leaves > CSSM_MAX_LEAVES) return fail(CSSM_EINVAL_DESC, "n_leaves = %d out of range", desc->n_leaves);
  int d = 0;
  for (int l = 0; l < desc->n_leaves; ++l) {
    const cssm_leaf_desc* L = &desc->leaves[l];
    if (L->dim < 1 || d + L->dim > CSSM_MAX_DIM) return fail(CSSM_EINVAL_DESC, "leaf %d: dimension %d (total > %d)", l, L->dim, CSSM_MAX_DIM);
    if (L->n_m0 < 1 || L->n_c0 < 1 || L->n_sigma < 1 || !L->m0 || !L->c0 || !L->sigma)
      return fail(CSSM_EINVAL_DESC, "leaf %d: m0, c0 and sigma are required", l);
    const bool need_mu = L->sde_kind == CSSM_SDE_GEN_BROWNIAN || L->sde_kind == CSSM_SDE_OU || L->sde_kind == CSSM_SDE_EULER_AFFINE;
    const bool need_phi = L->sde_kind == CSSM_SDE_OU || L->sde_kind == CSSM_SDE_EULER_AFFINE;
    if (L->sde_kind < 0 || L->sde_kind > CSSM_SDE_EULER_AFFINE) return fail(CSSM_EINVAL_DESC, "leaf %d: unknown sde_kind %d", l, L->sde_kind);
    if (need_mu && (L->n_mu < 1 || !L->mu)) return fail(CSSM_EINVAL_DESC, "leaf %d: mu is required", l);
    if (need_phi && (L->n_phi < 1 || !L->phi)) return fail(CSSM_EINVAL_DESC, "leaf %d: phi is required", l);
    if (L->f_kind == CSSM_F_SEASONAL) {
      if (L->dim != 2 * L->harmonics || L->period < 1) return fail(CSSM_EINVAL_DESC, "leaf %d: seasonal needs dim == 2*harmonics and period >= 1", l);
    } else if (L->f_kind != CSSM_F_FIRST) {
      return fail(CSSM_EINVAL_DESC, "leaf %d: unknown f_kind %d", l, L->f_kind);
    }
    for (int i = 0; i < L->dim; ++i) {
      Comp& c = pf->comp[d + i];
      c = Comp{};
      c.kind = L->sde_kind; c.leaf = l; c.idx = i; c.f_kind = L->f_kind; c.period = L->period;
      c.m0 = rep(L->m0, L->n_m0, i);
      c.c0 = cssm_exp(rep(L->c0, L->n_c0, i));
      switch (L->sde_kind) {
        case CSSM_SDE_BROWNIAN: c.sigma = cssm_exp(rep(L->sigma, L->n_sigma, i)); break;
        case CSSM_SDE_GEN_BROWNIAN: c.mu = rep(L->mu, L->n_mu, i); c.sigma = cssm_exp(rep(L->sigma, L->n_sigma, i)); break;
        case CSSM_SDE_OU:
          c.phi = logistic(rep(L->phi, L->n_phi, i));
          c.mu = rep(L->mu, L->n_mu, i);
          c.sigma = cssm_exp(rep(L->sigma, L->n_sigma, i));
          break;
        default: c.mu = rep(L->mu, L->n_mu, i); c.phi = rep(L->phi, L->n_phi, i); c.sigma = rep(L->sigma, L->n_sigma, i); break;
      }
    }
    d += L->dim;
  }
  pf->d = d;
  pf->n_leaves = desc->n_leaves;
  pf->obs_kind = desc->obs_kind;
  pf->precision = desc->lgcp_precision;
  switch (desc->obs_kind) {
    case CSSM_OBS_GAUSSIAN: case CSSM_OBS_NEGBIN: case CSSM_OBS_STUDENT_T: case CSSM_OBS_ZIP:
      // "Must provide SD parameter" / "No scale parameter provided", model/Model.scala:150,179,214,250,294
      if (!desc->leaves[0].has_scale) return fail(CSSM_EINVAL_DESC, "this observation model needs the scale parameter of the leftmost leaf");
      pf->scale_raw = desc->leaves[0].scale;
      pf->scale_sd = cssm_exp(desc->leaves[0].scale);  // model/Model.scala:147,171,244
      if (desc->obs_kind == CSSM_OBS_STUDENT_T && desc->obs_df < 1) return fail(CSSM_EINVAL_DESC, "Student-t needs obs_df >= 1");
      pf->obs_df = desc->obs_df;
      break;
    case CSSM_OBS_POISSON: case CSSM_OBS_LGCP: case CSSM_OBS_BERNOULLI: case CSSM_OBS_BETA: break;
    default: return fail(CSSM_EINVAL_DESC, "unknown obs_kind %d", desc->obs_kind);
  }
  if (desc->obs_kind >= CSSM_OBS_NEGBIN)   // these models take f = first component of EVERY leaf (Model.scala:153,184,296,328,347)
    for (int l = 0; l < desc->n_leaves; ++l)
      if (l == 0 && desc->leaves[l].f_kind != CSSM_F_FIRST) return fail(CSSM_EINVAL_DESC, "the observing leaf must use the first-component map");
  if (desc->obs_kind == CSSM_OBS_LGCP && (desc->lgcp_precision < 0 || desc->lgcp_precision > 9))
    return fail(CSSM_EINVAL_DESC, "lgcp_precision %d out of range", desc->lgcp_precision);
  // kernel-side constants
  ModelK& mk = pf->mk;
  memset(&mk, 0, sizeof mk);
  mk.d = d; mk.obs_kind = desc->obs_kind;
  for (int k = 0; k < d; ++k) {
    const Comp& c = pf->comp[k];
    uint32_t fm;
    if (c.f_kind == CSSM_F_FIRST) fm = (c.idx == 0) ? FM_START : FM_SKIP;
    else fm = (c.idx == 0) ? FM_START : FM_ADD;
    const uint32_t leaf_end = (k + 1 == d) || (pf->comp[k + 1].leaf != c.leaf);
    const uint32_t first_leaf = (c.leaf == 0);
    const uint32_t b = ((uint32_t)c.kind & 3u) | (fm << 2) | (leaf_end << 4) | (first_leaf << 5);
    mk.comp[k >> 2] |= b << ((k & 3) * 8);
  }
  return CSSM_OK;
}

static int upload_init_params(cssm_pf* pf) {
  double m0[CSSM_MAX_DIM], sd0[CSSM_MAX_DIM];
  for (int k = 0; k < pf->d; ++k) { m0[k] = pf->comp[k].m0; sd0[k] = std::sqrt(pf->comp[k].c0); }
  HIP_TRY(hipMemcpyAsync(pf->d_m0, m0, pf->d * 8, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(pf->d_sd0, sd0, pf->d * 8, hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  return CSSM_OK;
}

// Everything of one observation that does not depend on the particle: transition coefficients
// (model/Sde.scala:88-91,117-119,139-146), F(t) (model/Model.scala:217-223), the observation
// constants, the resampling uniform (model/Resampling.scala:66) and the sampleOne index (:152).
static void build_rec(const cssm_pf* pf, double t_prev, double t, double y, int has_obs, uint32_t step, StepRec* r) {
  memset(r, 0, sizeof *r);
  double dt = t - t_prev;                                      // model/ParticleFilter.scala:117
  r->has_obs = has_obs;
  r->step = step;
  r->n_sub = 0;
  r->t_obs = t;
  if (pf->obs_kind == CSSM_OBS_LGCP) {
    r->has_obs = 1;                                            // FilterLgcp always weights (:210-226)
    if (dt == 0) { r->n_sub = 0; }
    else {
      const double delta = std::pow(10.0, -pf->precision);     // :190
      r->n_sub = (int)std::ceil(dt / delta);
      dt = delta;
    }
  }
  r->dt = dt;
  for (int k = 0; k < pf->d; ++k) {
    const Comp& c = pf->comp[k];
    double* p = r->coef[k];
    switch (c.kind) {
      case CSSM_SDE_BROWNIAN: p[3] = std::sqrt(c.sigma * dt); break;
      case CSSM_SDE_GEN_BROWNIAN: p[0] = c.mu * dt; p[3] = std::sqrt(c.sigma * dt); break;
      case CSSM_SDE_OU: {
        const double var = (c.sigma * c.sigma / (c.phi * 2.0)) * (1.0 - cssm_exp(c.phi * -2.0 * dt));
        p[0] = c.mu; p[1] = cssm_exp(-c.phi * dt); p[3] = std::sqrt(var);
        break;
      }
      default: p[0] = c.mu; p[1] = c.phi; p[2] = c.sigma; p[3] = std::sqrt(dt); break;
    }
    if (c.f_kind == CSSM_F_FIRST) {
      r->fco[k] = (c.idx == 0) ? 1.0 : 0.0;
    } else {
      double sn, cs;
      cssm_sincos2pi(cssm_seasonal_phase((double)(c.idx / 2 + 1), t, (double)c.period), &sn, &cs);
      r->fco[k] = (c.idx & 1) ? sn : cs;
    }
  }
  const long long k = (long long)y;                            // y.toInt
  r->y = y;
  switch (pf->obs_kind) {
    case CSSM_OBS_POISSON:                                     // c0 = lgamma(k+1)
      r->y = (double)k; r->c[0] = cssm_lgamma_kp1(k); break;
    case CSSM_OBS_GAUSSIAN:                                    // c0 = log(sqrt(2 pi) sd), c1 = sd
      r->c[0] = cssm_log(2.5066282746310002 * pf->scale_sd); r->c[1] = pf->scale_sd; break;
    case CSSM_OBS_NEGBIN: {                                    // c0 = lgamma(size+k) - lgamma(k+1) - lgamma(size), c1 = size
      const double size = pf->scale_sd;
      r->y = (double)k;
      r->c[0] = cssm_lgamma(size + (double)k) - cssm_lgamma_kp1(k) - cssm_lgamma(size); r->c[1] = size;
      break;
    }
    case CSSM_OBS_ZIP: {                                       // c0 = p, c1 = -log(1 + exp(v)), c2 = lgamma(k+1)
      const double ev = cssm_exp(pf->scale_raw);
      r->y = (double)k;
      r->c[0] = ev / (1.0 + ev); r->c[1] = -cssm_log(1.0 + ev); r->c[2] = cssm_lgamma_kp1(k);
      break;
    }
    case CSSM_OBS_STUDENT_T: {                                 // c0 = -logNormalizer, c1 = v, c2 = (df+1)/2, c3 = 1/v
      const double df = (double)pf->obs_df, v = pf->scale_sd;
      r->c[0] = cssm_lgamma((df + 1.0) / 2.0) - cssm_lgamma(df / 2.0) - 0.5 * cssm_log(3.14159265358979311600 * df);
      r->c[1] = v; r->c[2] = (df + 1.0) / 2.0; r->c[3] = 1.0 / v; r->cdf = df;
      break;
    }
    case CSSM_OBS_BETA: r->c[0] = cssm_log(y); break;          // c0 = log(y)
    default: break;                                            // Bernoulli, LGCP: no constants
  }
  // reference level of the step's weights (include/cssm_numerics.h); NaN = rescale by the max
  r->ref = (pf->obs_kind == CSSM_OBS_LGCP) ? cssm_nan() : cssm_ref_level(pf->obs_kind, y, pf->scale_sd, (double)pf->obs_df);
  const cssm_u32x4 bu = cssm_philox_draw(pf->seed, 0, step, CSSM_STREAM_U, 0);
  r->u = cssm_u01(bu.v[0], bu.v[1]);
  const int32_t pr = (int32_t)cssm_philox_draw(pf->seed, 0, step + 1, CSSM_STREAM_PICK, 0).v[0];
  const uint32_t pa = pr < 0 ? (uint32_t)0 - (uint32_t)pr : (uint32_t)pr;
  r->pick = (uint32_t)((uint64_t)pa % pf->n_global);
}

// FilterLgcp.calcWeight evaluates f at EVERY simulated time tau_s = t + s delta, the clock starting at the observation's
// time (model/ParticleFilter.scala:193-205, :215; model/Sde.scala:57-66 -- a reference quirk that only shows when f depends
// on time, i.e. with a seasonal leaf).  For such models the coefficients c_k(tau_s) of records [first, first + count) are
// tabulated here (they depend on (t, s) only; tau is accumulated by repeated addition exactly as the oracle does) and
// uploaded behind the records; the kernel reads row s of its observation.  `reset`: the records start a new table.
static int build_fsub(cssm_pf* pf, size_t first, size_t count, bool reset) {
  if (!pf->lgcp_tdep) return CSSM_OK;
  if (reset) pf->h_fsub.clear();
  const int d = pf->d;
  for (size_t q = first; q < first + count; ++q) {
    StepRec* r = &pf->h_recs[q];
    r->fsub_off = 0;
    if (r->n_sub <= 0) continue;
    if (pf->h_fsub.size() + (size_t)r->n_sub * d > ((size_t)1 << 27))
      return fail(CSSM_ENOMEM, "the sub-step table of a time-dependent LGCP model would exceed 1 GiB (%d sub-steps at observation %zu)", r->n_sub, q);
    r->fsub_off = (uint32_t)pf->h_fsub.size();
    double tau = r->t_obs;
    for (int sidx = 0; sidx < r->n_sub; ++sidx) {
      tau = tau + r->dt;                                   // t = s.time + dt, model/Sde.scala:60 (r->dt is delta for LGCP)
      for (int k = 0; k < d; ++k) {
        const Comp& c = pf->comp[k];
        double v;
        if (c.f_kind == CSSM_F_FIRST) v = (c.idx == 0) ? 1.0 : 0.0;
        else {
          double sn, cs;
          cssm_sincos2pi(cssm_seasonal_phase((double)(c.idx / 2 + 1), tau, (double)c.period), &sn, &cs);
          v = (c.idx & 1) ? sn : cs;
        }
        pf->h_fsub.push_back(v);
      }
    }
  }
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
static int prop_items(int d) { return d <= 2 ? CSSM_PROP_IT_LO : (d <= 8 ? CSSM_PROP_IT_MID : 1); }   // PropItems<D>

// k_propagate blocks per 1024-particle unit on a single-GPU handle: one tile of the kernel per block below 2^20 particles
static uint32_t auto_split(const cssm_pf* pf) {
  if (pf->sharded || pf->sup != 1 || pf->n >= CSSM_SPLIT_MAX_N) return 1u;
  return prop_items(pf->d) == 1 ? 4u : 2u;   // one tile of the kernel: half of 1024 (two particles per thread), a quarter (one)
}

static int alloc_handle(cssm_pf* pf) {
  HIP_TRY(hipSetDevice(pf->device));
  if (pf->own_stream) HIP_TRY(hipStreamCreateWithFlags(&pf->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&pf->ev0));
  HIP_TRY(hipEventCreate(&pf->ev1));
  pf->stride = (size_t)((pf->n + CSSM_TILE - 1) / CSSM_TILE) * CSSM_TILE;   // rows start 16-B aligned
  pf->ntiles = (uint32_t)((pf->n + CSSM_TILE - 1) / CSSM_TILE);
  pf->sup = (pf->ntiles + 1023u) / 1024u;
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
  if (sharded) { pf->stream = (hipStream_t)stream; pf->own_stream = false; pf->opt_fused = 1; }
  else pf->opt_fused = 1;   // two launches per observation at every size (measured with the slim single-GPU kernels: 18.7 vs 20.0 us at
                            // N = 100 000, 34.0 vs 35.4 at 2^20, 336 vs 356 at 2^24); an outlying observation is redone in place
  int rc = build_model(pf, desc, false);
  if (rc == CSSM_OK) rc = alloc_handle(pf);
  if (rc != CSSM_OK) { std::string keep = g_err; cssm_pf_destroy(pf); g_err = keep; return rc; }
  *out = pf;
  return CSSM_OK;
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
  void* ptrs[] = {pf->fineS, pf->fineS2, pf->logw_alt, pf->tileS_alt, pf->tileS2_alt, pf->state[0], pf->state[1], pf->logw, pf->endslot, pf->anc, pf->tileS, pf->tileS2, pf->tileP, pf->sc,
                  pf->d_m0, pf->d_sd0, pf->d_logtab, pf->d_sync, pf->d_ts, pf->d_fsub, pf->cum, pf->d_recs, pf->d_ll_t, pf->d_ess_t, pf->d_path, pf->cand, pf->cand_end, pf->cand_idx, pf->d_bounds, pf->d_xch, pf->d_need};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (pf->h_recs) (void)hipHostFree(pf->h_recs);
  if (pf->h_sc) (void)hipHostFree(pf->h_sc);
  for (hipEvent_t e : pf->prof_ev) (void)hipEventDestroy(e);
  if (pf->ev0) (void)hipEventDestroy(pf->ev0);
  if (pf->ev1) (void)hipEventDestroy(pf->ev1);
  if (pf->own_stream && pf->stream) (void)hipStreamDestroy(pf->stream);
  delete pf;
}

extern "C" int cssm_pf_set_params(cssm_pf* pf, const cssm_model_desc* desc) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = build_model(pf, desc, true);
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

static const int kGridCap = 4096;

// ll = 0.0, ess = N: PfState(t0, None, state, 0.0, particles), model/ParticleFilter.scala:107
static int reset_scalars(cssm_pf* pf) {
  Scalars h;
  memset(&h, 0, sizeof h);
  h.ess = (int32_t)(pf->n_global < 2147483647ull ? pf->n_global : 2147483647ull);
  h.fail_step = 0xffffffffu;
  // pageable source: the copy is staged before the call returns, so a stack object is safe
  HIP_TRY(hipMemcpyAsync(pf->sc, &h, sizeof h, hipMemcpyHostToDevice, pf->stream));
  return CSSM_OK;
}

static int launch_init(cssm_pf* pf, double t0) {
  HIP_TRY(hipSetDevice(pf->device));
  const int grid = grid_for(pf->n, CSSM_BLOCK, kGridCap);
  DISPATCH_D(pf->d, k_init<D><<<dim3(grid), dim3(CSSM_BLOCK), 0, pf->stream>>>(pf->state[0], pf->stride, pf->n, pf->first,
                                                                              pf->seed, pf->d_m0, pf->d_sd0, pf->d_logtab));
  HIP_TRY(hipGetLastError());
  int rc = reset_scalars(pf);
  if (rc) return rc;
  pf->cur = 0; pf->src = pf->state[0]; pf->src_stride = pf->stride; pf->anc_valid = false;
  pf->t = t0; pf->step = 0; pf->initialised = true; pf->wparity = 0;
  return CSSM_OK;
}

// propagate + weight of one datum (record already on the device)

// whether launch_propagate will use the kernels that also form the sums (and can record sampleOne's pick on the way)
static bool uses_sums_kernel(const cssm_pf* pf) {
  return pf->opt_fused && !pf->safe_sums && pf->obs_kind != CSSM_OBS_LGCP && pf->resampler != CSSM_RESAMPLE_MULTINOMIAL;
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
  if (pf->opt_whole == 3) return pf->no_fine ? GEO_LOOP : GEO_FINE;
  const uint32_t tiles = pf->sup * (uint32_t)CSSM_TILE / (uint32_t)(CSSM_BLOCK * prop_items(pf->d));
  if (tiles <= CSSM_LOOP_MAX_TILES) return GEO_LOOP;
  return (pf->d >= CSSM_FINE_MIN_D && !pf->no_fine) ? GEO_FINE : GEO_PIPE;
}

static int launch_propagate(cssm_pf* pf, const StepRec* d_rec, double* pick_out = nullptr, uint32_t pick_slot = 0) {
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
  if (fine) chunk = (uint64_t)CSSM_BLOCK * prop_items(pf->d);   // (divides the unit: 1024 * sup)
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
  a.grid = grid; a.stream = pf->stream;
  a.lgcp = pf->obs_kind == CSSM_OBS_LGCP;
  a.obs = (pf->obs_kind == CSSM_OBS_POISSON || pf->obs_kind == CSSM_OBS_GAUSSIAN) ? pf->obs_kind : -1;
  a.sums = do_sums;
  a.src = pf->src; a.src_stride = pf->src_stride; a.anc = anc; a.dst = dst; a.dst_stride = pf->stride; a.logw = pf->logw;
  a.n = pf->n; a.gid0 = pf->first; a.seed = pf->seed; a.rec = d_rec; a.mk = pf->mk; a.sc = pf->sc;
  a.slot_set = pf->sharded ? 0 : pf->wparity;
  a.src2 = anc ? pf->src2 : nullptr; a.src2_stride = pf->src2_stride; a.n_split = pf->n_split; a.logtab = pf->d_logtab;
  a.chunk = chunk; a.do_sums = do_sums; a.subS = fine ? pf->fineS : pf->tileS; a.subS2 = fine ? pf->fineS2 : pf->tileS2;
  a.pick_out = pick_out; a.pick_slot = pick_slot;
  a.fsub = pf->lgcp_tdep ? pf->d_fsub : nullptr;
  a.one = (chunk == (uint64_t)CSSM_BLOCK * prop_items(pf->d)) ? 1 : (geo == GEO_LOOP ? 2 : 0);
  // sharded handle on the single-collective exchange (received rows read in place: src2_stride == 0) or before its first exchange:
  // slim launch, tile after tile while a unit has at most CSSM_LOOP_MAX_TILES tiles; whole pairs per thread (d <= 8) need an even first id
  a.shard_slim = pf->sharded && do_sums && !a.lgcp && (a.src2 == nullptr || a.src2_stride == 0) && a.fsub == nullptr && pick_out == nullptr &&
                 ((pf->first & 1ull) == 0ull || prop_items(pf->d) == 1) && a.slot_set == 0;
  if (a.shard_slim && pf->sup * (uint32_t)CSSM_TILE / (uint32_t)(CSSM_BLOCK * prop_items(pf->d)) <= CSSM_LOOP_MAX_TILES) a.one = 2;
  switch (pf->d) {
#define CSSM_CASE_PROP(D) case D: cssm_prop_launch_d##D(a); break;
    CSSM_CASE_PROP(1) CSSM_CASE_PROP(2) CSSM_CASE_PROP(3) CSSM_CASE_PROP(4) CSSM_CASE_PROP(5) CSSM_CASE_PROP(6) CSSM_CASE_PROP(7) CSSM_CASE_PROP(8)
    CSSM_CASE_PROP(9) CSSM_CASE_PROP(10) CSSM_CASE_PROP(11) CSSM_CASE_PROP(12) CSSM_CASE_PROP(13) CSSM_CASE_PROP(14) CSSM_CASE_PROP(15)
    default: cssm_prop_launch_d16(a); break;
#undef CSSM_CASE_PROP
  }
  prof_end(pf);
  if (fine) {   // the blocks' sums -> the units' (the kernel itself skips unweighted observations and series on hold)
    prof_begin(pf, CSSM_K_REDUCE);
    hipLaunchKernelGGL(k_reduce_units, dim3((pf->nunits + CSSM_BLOCK / 64 - 1) / (CSSM_BLOCK / 64)), dim3(CSSM_BLOCK), 0, pf->stream,
                       (const cssm_u128*)pf->fineS, (const cssm_u128*)pf->fineS2, (uint32_t)grid, (uint32_t)(unit_particles / chunk), pf->nunits,
                       pf->tileS, pf->tileS2, (const Scalars*)pf->sc, d_rec);
    prof_end(pf);
  }
  HIP_TRY(hipGetLastError());
  pf->cur ^= 1;
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->anc_valid = false; pf->src2 = nullptr;
  return CSSM_OK;
}

// sums (unless k_propagate formed them) -> end slots -> ancestors, single GPU.  `redo`: second attempt at the step
// whose reference level the max ruled out -- same max-slot set, sums formed by k_tile_sums with the agreed level.
static int launch_resample(cssm_pf* pf, const StepRec* d_rec, double* ll_t = nullptr, int32_t* ess_t = nullptr, uint32_t rec_idx = 0,
                           bool redo = false) {
  if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL && !pf->cum) {
    if (hipMalloc(&pf->cum, pf->stride * 8) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc cumulative weights");
  }
  if (redo) pf->wparity = (pf->wparity + CSSM_MAXSETS - 1) % CSSM_MAXSETS;
  const bool optimistic = pf->last_optimistic && !redo;
  const int tgrid = (int)pf->nunits;
  const int split = optimistic ? (int)pf->split : 1;
  if (!optimistic) {
    prof_begin(pf, CSSM_K_TILE_SUMS);
    hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc, pf->tileS, pf->tileS2, pf->ntiles,
                       pf->sup, pf->nunits, 0, pf->wparity, (const double*)nullptr, pf->d_logtab, d_rec);
    prof_end(pf);
  }
  prof_begin(pf, CSSM_K_OFFSPRING);   // unit prefix, ll/ess, end slots and their expansion to ancestors in one kernel (one block per unit + the publisher)
#define OFF_ARGS pf->logw, pf->n, pf->sc, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, d_rec, pf->anc, pf->ntiles, pf->sup, pf->nunits, \
                 pf->wparity, ll_t, ess_t, rec_idx, pf->opt_exact, split, pf->seed, pf->cum, optimistic ? (pf->batch_hold ? 3 : 1) : 0
  if (pf->resampler == CSSM_RESAMPLE_STRATIFIED)
    hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_STRATIFIED>), dim3(tgrid + 1), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS);
  else if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL)
    hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_MULTINOMIAL>), dim3(tgrid + 1), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS);
  else
    hipLaunchKernelGGL((k_offspring_self<CSSM_RESAMPLE_SYSTEMATIC>), dim3(tgrid + 1), dim3(CSSM_BLOCK), 0, pf->stream, OFF_ARGS);
#undef OFF_ARGS
  if (pf->resampler == CSSM_RESAMPLE_MULTINOMIAL)
    hipLaunchKernelGGL(k_multinomial, dim3(grid_for(pf->n, 256, kGridCap)), dim3(256), 0, pf->stream, pf->cum, pf->n, pf->seed,
                       pf->h_step_for_resample, pf->anc);
  prof_end(pf);
  pf->wparity = (pf->wparity + 1) % CSSM_MAXSETS;
  HIP_TRY(hipGetLastError());
  pf->anc_valid = true;
  return CSSM_OK;
}

// ---- one launch per observation (k_step, cssm_propagate.hip.h) ---------------------------------------------------------
#ifndef CSSM_STEP_MAX_N
#define CSSM_STEP_MAX_N (1u << 18)   /* CSSM_OPT_ONE_LAUNCH = -1: up to here the one-launch path is within 5 % of the two-launch path */
#endif
static void swap_sets(cssm_pf* pf) {
  std::swap(pf->logw, pf->logw_alt); std::swap(pf->tileS, pf->tileS_alt); std::swap(pf->tileS2, pf->tileS2_alt);
  pf->pp ^= 1;
}
// whether the batch drivers may merge the resampling of one observation with the propagate of the next
static bool step_eligible(const cssm_pf* pf) {
  if (pf->opt_step == 0 || !uses_sums_kernel(pf) || pf->sharded || pf->resampler != CSSM_RESAMPLE_SYSTEMATIC) return false;
  if (pf->first != 0 || pf->n != pf->n_global || pf->sup != 1) return false;
  if (pf->split != 1 && (uint32_t)CSSM_TILE / pf->split != (uint32_t)(CSSM_BLOCK * prop_items(pf->d))) return false;   // whole tiles, or one tile of the kernel
  if ((pf->n + CSSM_TILE / pf->split - 1) / (CSSM_TILE / pf->split) > CSSM_STEP_UNITS) return false;   // the unit sums a block scans
  return pf->opt_step > 0 || pf->n <= CSSM_STEP_MAX_N;
}
static int ensure_step_sets(cssm_pf* pf) {
  if (pf->logw_alt) return CSSM_OK;
  const size_t nsums = (size_t)(pf->ntiles > 4 * pf->nunits ? pf->ntiles : 4 * pf->nunits);
  if (hipMalloc(&pf->logw_alt, pf->stride * 8) != hipSuccess) return fail(CSSM_ENOMEM, "hipMalloc of the second set of log-weights");
  HIP_TRY(hipMalloc(&pf->tileS_alt, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMalloc(&pf->tileS2_alt, nsums * sizeof(cssm_u128)));
  HIP_TRY(hipMemsetAsync(pf->tileS_alt, 0, nsums * sizeof(cssm_u128), pf->stream));
  HIP_TRY(hipMemsetAsync(pf->tileS2_alt, 0, nsums * sizeof(cssm_u128), pf->stream));
  return CSSM_OK;
}
// resampling of the observation before d_rec (not launched by its own step: deferred) + propagate and weight of d_rec
static int launch_merged_step(cssm_pf* pf, const StepRec* d_rec, double* ll_t, int32_t* ess_t, double* pick_out, uint32_t pick_slot) {
  int rc = ensure_step_sets(pf);
  if (rc) return rc;
  StepLaunch a;
  a.chunk = (uint32_t)CSSM_TILE / pf->split;
  a.grid = (int)((pf->n + a.chunk - 1) / a.chunk); a.stream = pf->stream;
  a.obs = (pf->obs_kind == CSSM_OBS_POISSON || pf->obs_kind == CSSM_OBS_GAUSSIAN) ? pf->obs_kind : -1;
  a.src = pf->src; a.src_stride = pf->src_stride; a.dst = pf->state[pf->cur ^ 1]; a.dst_stride = pf->stride;
  a.logw_in = pf->logw; a.logw_out = pf->logw_alt; a.n = (uint32_t)pf->n; a.seed = pf->seed; a.rec = d_rec; a.mk = pf->mk; a.sc = pf->sc;
  a.set_in = pf->wparity; a.logtab = pf->d_logtab;
  a.inS = pf->tileS; a.inS2 = pf->tileS2; a.outS = pf->tileS_alt; a.outS2 = pf->tileS2_alt; a.nunits = (uint32_t)a.grid;
  a.ll_t = ll_t; a.ess_t = ess_t; a.force_exact = pf->opt_exact; a.pick_out = pick_out; a.pick_slot = pick_slot;
  prof_begin(pf, CSSM_K_STEP);
  switch (pf->d) {
#define CSSM_CASE_STEP(D) case D: cssm_step_launch_d##D(a); break;
    CSSM_CASE_STEP(1) CSSM_CASE_STEP(2) CSSM_CASE_STEP(3) CSSM_CASE_STEP(4) CSSM_CASE_STEP(5) CSSM_CASE_STEP(6) CSSM_CASE_STEP(7) CSSM_CASE_STEP(8)
    CSSM_CASE_STEP(9) CSSM_CASE_STEP(10) CSSM_CASE_STEP(11) CSSM_CASE_STEP(12) CSSM_CASE_STEP(13) CSSM_CASE_STEP(14) CSSM_CASE_STEP(15)
    default: cssm_step_launch_d16(a); break;
#undef CSSM_CASE_STEP
  }
  prof_end(pf);
  HIP_TRY(hipGetLastError());
  swap_sets(pf);
  pf->wparity = (pf->wparity + 1) % CSSM_MAXSETS;
  pf->cur ^= 1;
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->anc_valid = false; pf->src2 = nullptr;
  pf->last_optimistic = true;
  return CSSM_OK;
}

static int launch_step(cssm_pf* pf, const StepRec* d_rec, int weighted, uint32_t step_index, double* ll_t = nullptr, int32_t* ess_t = nullptr,
                       uint32_t rec_idx = 0, double* pick_out = nullptr, uint32_t pick_slot = 0) {
  pf->h_step_for_resample = step_index;
  int rc = launch_propagate(pf, d_rec, pick_out, pick_slot);
  if (rc) return rc;
  if (weighted) rc = launch_resample(pf, d_rec, ll_t, ess_t, rec_idx);
  else if (ll_t) hipLaunchKernelGGL(k_record, dim3(1), dim3(1), 0, pf->stream, pf->sc, ll_t, ess_t, rec_idx);
  return rc;
}

// The host reads the scalars BEHIND the max slots (err, ess, fail_step, gmax, ref, ll, the sums: ~120 bytes, not the 24 KiB of
// slot lines in front of them -- that copy to pageable memory cost ~15 us per streaming step and per batch call).
#define CSSM_SC_TAIL_OFF offsetof(Scalars, err)
#define CSSM_SC_TAIL_ARGS(hp, scp) reinterpret_cast<char*>(hp) + CSSM_SC_TAIL_OFF, reinterpret_cast<const char*>(scp) + CSSM_SC_TAIL_OFF, sizeof(Scalars) - CSSM_SC_TAIL_OFF

static int check_device_err(cssm_pf* pf, const Scalars& h) {
  if (h.err & 1u) return fail(CSSM_ENONFINITE, "a log-weight is NaN (the reference's breeze distribution constructor would throw)");
  if (h.err & 2u) return fail(CSSM_ENONFINITE, "all particle weights are zero or the maximum log-weight is not finite");
  (void)pf;
  return CSSM_OK;
}

// ------------------------------------------------------------------------------------ streaming API

static int ensure_recs(cssm_pf* pf, size_t T) {
  if (T < 1024) T = 1024;   // one allocation serves every ordinary series length
  if (pf->h_recs_cap < T) {
    if (pf->h_recs) (void)hipHostFree(pf->h_recs);
    pf->h_recs = nullptr;
    HIP_TRY(hipHostMalloc((void**)&pf->h_recs, T * sizeof(StepRec), hipHostMallocDefault));
    pf->h_recs_cap = T;
  }
  if (pf->recs_cap < T) {
    if (pf->d_recs) (void)hipFree(pf->d_recs);
    if (pf->d_ll_t) (void)hipFree(pf->d_ll_t);
    if (pf->d_ess_t) (void)hipFree(pf->d_ess_t);
    pf->d_recs = nullptr; pf->d_ll_t = nullptr; pf->d_ess_t = nullptr;
    HIP_TRY(hipMalloc(&pf->d_recs, T * sizeof(StepRec)));
    HIP_TRY(hipMalloc(&pf->d_ll_t, T * 8));
    HIP_TRY(hipMalloc(&pf->d_ess_t, T * 4));
    pf->recs_cap = T;
  }
  return CSSM_OK;
}

extern "C" int cssm_pf_init(cssm_pf* pf, double t0) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  int rc = launch_init(pf, t0);
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
  pf->cur = 0; pf->src = pf->state[0]; pf->src_stride = pf->stride; pf->anc_valid = false;
  pf->t = t0; pf->step = 0; pf->initialised = true; pf->wparity = 0;
  return CSSM_OK;
}

extern "C" int cssm_pf_step(cssm_pf* pf, double t, double obs, int has_obs, double* ll_out, int32_t* ess_out) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (!pf->initialised) return fail(CSSM_ESTATE, "cssm_pf_step before cssm_pf_init");
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = ensure_recs(pf, 1);
  if (rc) return rc;
  build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[0]);
  rc = build_fsub(pf, 0, 1, true);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pf->d_recs, pf->h_recs, sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  const int weighted = pf->h_recs[0].has_obs;
  rc = launch_step(pf, pf->d_recs, weighted, pf->step);
  if (rc) return rc;
  if (!pf->h_sc) HIP_TRY(hipHostMalloc((void**)&pf->h_sc, sizeof(Scalars), hipHostMallocDefault));
  Scalars& h = *pf->h_sc;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (h.err == 4u) {   // the max ruled the reference level out: the log-weights are in place, form the sums again
    HIP_TRY(hipMemsetAsync(&pf->sc->err, 0, sizeof(uint32_t), pf->stream));
    rc = launch_resample(pf, pf->d_recs, nullptr, nullptr, 0, /*redo=*/true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
    HIP_TRY(hipStreamSynchronize(pf->stream));
  }
  pf->t = t; pf->step++;
  if (ll_out) *ll_out = h.ll;
  if (ess_out) *ess_out = h.ess;
  return check_device_err(pf, h);
}

// ------------------------------------------------------------------------------------ persistent series kernel

static int series_items(int d) { return d <= 8 ? 2 : 1; }   // SeriesItems<D>

#define CSSM_SER_DISPATCH(fn, d, ...)                                                                       \
  switch (d) {                                                                                               \
    case 1: e = fn##1(__VA_ARGS__); break;   case 2: e = fn##2(__VA_ARGS__); break;   case 3: e = fn##3(__VA_ARGS__); break;   \
    case 4: e = fn##4(__VA_ARGS__); break;   case 5: e = fn##5(__VA_ARGS__); break;   case 6: e = fn##6(__VA_ARGS__); break;   \
    case 7: e = fn##7(__VA_ARGS__); break;   case 8: e = fn##8(__VA_ARGS__); break;   case 9: e = fn##9(__VA_ARGS__); break;   \
    case 10: e = fn##10(__VA_ARGS__); break; case 11: e = fn##11(__VA_ARGS__); break; case 12: e = fn##12(__VA_ARGS__); break; \
    case 13: e = fn##13(__VA_ARGS__); break; case 14: e = fn##14(__VA_ARGS__); break; case 15: e = fn##15(__VA_ARGS__); break; \
    default: e = fn##16(__VA_ARGS__); break;                                                                 \
  }

// How the series kernel would run this handle: `grid` blocks of `per_block` consecutive particles each, or not at all
// (false): LGCP, other resamplers and sharded handles use the per-observation kernels; a block must be able to keep its
// particles' log-weights in LDS (per_block <= CSSM_SER_LW_CAP); all blocks must be resident together.
struct SeriesPlan { int grid; uint32_t per_block; size_t smem; };
static bool series_plan(cssm_pf* pf, SeriesPlan* plan) {
  if (!pf->opt_series || pf->sharded || pf->obs_kind == CSSM_OBS_LGCP || pf->resampler != CSSM_RESAMPLE_SYSTEMATIC) return false;
  if (pf->first != 0 || pf->n != pf->n_global || pf->n < 1) return false;
  const int obs = (pf->obs_kind == CSSM_OBS_POISSON || pf->obs_kind == CSSM_OBS_GAUSSIAN) ? pf->obs_kind : -1;
  const uint64_t tile = (uint64_t)CSSM_BLOCK * series_items(pf->d);
  if (pf->ser_blocks_max < 0) {   // asked once per handle: can this device launch cooperatively at all
    pf->ser_blocks_max = 0;
    int cus = 0, coop = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pf->device) == hipSuccess &&
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, pf->device) == hipSuccess && coop && cus > 0)
      pf->ser_blocks_max = cus;   // (holds the CU count; blocks per CU depend on the LDS a launch asks for, below)
    (void)hipGetLastError();
  }
  if (pf->ser_blocks_max < 1) return false;
  // the fewest particles per block that CSSM_SER_MAXBLOCKS blocks can cover, then upwards until the blocks that size needs
  // are resident together (more log-weights in LDS per block = fewer blocks per CU)
  uint64_t per = (pf->n + CSSM_SER_MAXBLOCKS - 1) / CSSM_SER_MAXBLOCKS;
  per = (per + tile - 1) / tile * tile;
  for (; per <= CSSM_SER_LW_CAP; per += tile) {
    const size_t idx = (size_t)(per / tile);
    if (pf->ser_occ.size() <= idx) pf->ser_occ.resize(idx + 1, -1);
    if (pf->ser_occ[idx] < 0) {
      int per_cu = 0;
      hipError_t e = hipSuccess;
      CSSM_SER_DISPATCH(cssm_series_occupancy_d, pf->d, obs, (size_t)per * 8, &per_cu);
      pf->ser_occ[idx] = (e == hipSuccess && per_cu > 0) ? per_cu : 0;
      (void)hipGetLastError();
    }
    long long resident = (long long)pf->ser_occ[idx] * pf->ser_blocks_max;
    if (resident > CSSM_SER_MAXBLOCKS) resident = CSSM_SER_MAXBLOCKS;
    if ((long long)((pf->n + per - 1) / per) <= resident) break;
  }
  if (per > CSSM_SER_LW_CAP) return false;
  plan->per_block = (uint32_t)per;
  plan->grid = (int)((pf->n + per - 1) / per);
  plan->smem = (size_t)per * 8;
  return true;
}

// One cooperative launch for observations [0, T) (records already on the device, cloud initialised in state[0]).
static int launch_series(cssm_pf* pf, const SeriesPlan& plan, size_t T, double* d_path) {
  if (!pf->d_sync) HIP_TRY(hipMalloc(&pf->d_sync, cssm_series_sync_bytes()));
  HIP_TRY(hipMemsetAsync(pf->d_sync, 0, cssm_series_sync_bytes(), pf->stream));
  unsigned long long* ts = nullptr;
  const uint32_t ts_blocks = (pf->profile_level >= 2) ? (uint32_t)plan.grid : 1u;
  if (pf->profile) {
    const size_t need = (size_t)ts_blocks * T;
    if (pf->ts_cap < need) {
      if (pf->d_ts) (void)hipFree(pf->d_ts);
      pf->d_ts = nullptr; pf->ts_cap = 0;
      HIP_TRY(hipMalloc(&pf->d_ts, need * CSSM_SER_TS_PER_STEP * 8));
      pf->ts_cap = need;
    }
    ts = pf->d_ts;
  }
  pf->ts_blocks = ts_blocks; pf->ts_T = T;
  SeriesLaunch a;
  a.grid = plan.grid; a.stream = pf->stream; a.smem = plan.smem;
  a.obs = (pf->obs_kind == CSSM_OBS_POISSON || pf->obs_kind == CSSM_OBS_GAUSSIAN) ? pf->obs_kind : -1;
  a.state0 = pf->state[0]; a.state1 = pf->state[1]; a.stride = pf->stride; a.anc = pf->anc; a.logw = pf->logw;
  a.n = pf->n; a.seed = pf->seed; a.recs = pf->d_recs; a.T = (uint32_t)T; a.mk = pf->mk; a.sc = pf->sc; a.sync = pf->d_sync;
  a.logtab = pf->d_logtab; a.per_block = plan.per_block; a.cur0 = pf->cur; a.force_exact = pf->opt_exact;
  a.ll_t = pf->d_ll_t; a.ess_t = pf->d_ess_t; a.path = d_path; a.ts = ts; a.ts_blocks = ts_blocks;
  prof_begin(pf, CSSM_K_SERIES);
  hipError_t e = hipSuccess;
  CSSM_SER_DISPATCH(cssm_series_launch_d, pf->d, a);
  prof_end(pf);
  if (e != hipSuccess) return fail(CSSM_EHIP, "cooperative launch of the series kernel (%d blocks): %s", plan.grid, hipGetErrorString(e));
  // host-side mirror of what the kernel did: T propagates, the last resampling (if any) valid
  pf->cur = (pf->cur + (int)(T & 1)) & 1;
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->src2 = nullptr;
  pf->anc_valid = pf->h_recs[T - 1].has_obs != 0;
  pf->last_optimistic = true;
  return CSSM_OK;
}

// profiling: block 0's timestamps -> average duration of the four stretches of a weighted observation
static int series_collect_phases(cssm_pf* pf, size_t T) {
  std::vector<unsigned long long> h(T * CSSM_SER_TS_PER_STEP);
  HIP_TRY(hipMemcpy(h.data(), pf->d_ts, h.size() * 8, hipMemcpyDeviceToHost));
  double acc[4] = {0, 0, 0, 0};
  uint64_t cnt = 0;
  for (size_t s = 0; s < T; ++s) {
    if (!pf->h_recs[s].has_obs) continue;
    const unsigned long long* q = &h[s * CSSM_SER_TS_PER_STEP];
    for (int k = 0; k < 4; ++k) acc[k] += (double)(q[k + 1] - q[k]) * 0.01;   // 100 MHz ticks -> us
    ++cnt;
  }
  for (int k = 0; k < 4; ++k) pf->ser_phase_us[k] = cnt ? acc[k] / (double)cnt : 0.0;
  pf->ser_phase_steps = cnt;
  return CSSM_OK;
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
  int rc = ensure_recs(pf, 1);
  if (rc) return rc;
  build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[0]);
  rc = build_fsub(pf, 0, 1, true);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pf->d_recs, pf->h_recs, sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  rc = launch_propagate(pf, pf->d_recs);
  if (rc) return rc;
  // nobody decodes this step's running max on the device: clear both slot sets for the next weighted step
  HIP_TRY(hipMemsetAsync(pf->sc->maxslot, 0, sizeof(pf->sc->maxslot), pf->stream));
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  pf->wparity = 0;
  pf->t = t; pf->step++;
  return check_device_err(pf, h);
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
  if (pf->sharded) return fail(CSSM_ESTATE, "a sharded handle is driven through the cssm_pf_shard_* stages");
  HIP_TRY(hipSetDevice(pf->device));
  int rc = ensure_recs(pf, T);
  if (rc) return rc;
  if (cont && (!pf->initialised || path)) return fail(CSSM_ESTATE, "continuing a filter needs an initialised handle (and records no path)");
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];     // data.minBy(_.t).t, model/ParticleFilter.scala:138
  const uint32_t base = cont ? pf->step : 0u;                  // index of this call's first observation in the filter's series
  double tp = cont ? pf->t : t0;
  for (size_t s = 0; s < T; ++s) { build_rec(pf, tp, t[s], y[s], has ? has[s] : 1, base + (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = build_fsub(pf, 0, T, true);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pf->d_recs, pf->h_recs, T * sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  if (!cont) {
    rc = launch_init(pf, t0);
    if (rc) return rc;
  }
  // host-side state at the first observation of this call (launch_init: cur = 0, wparity = 0)
  const int cur0 = pf->cur, wpar0 = pf->wparity;
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
  HIP_TRY(hipEventRecord(pf->ev0, pf->stream));
  SeriesPlan plan;
  pf->last_series = !cont && series_plan(pf, &plan);
  if (pf->last_series) {
    // all T observations in one cooperative launch (cssm_series.hip.h); path entries 1 .. T-1 are recorded inside it, the
    // last one (no following propagate) by k_pick
    rc = launch_series(pf, plan, T, path ? pf->d_path : nullptr);
    if (rc) return rc;
    if (path)
      hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride,
                         (const uint32_t*)(pf->anc_valid ? pf->anc : nullptr), (uint64_t)pf->h_recs[T - 1].pick, d,
                         pf->d_path + T * (size_t)d);
  }
  // path entry s + 1 = the resampled state sampleOne picks after observation s.  With the kernels that also form the
  // sums (small handles: the PMMH case) the k_propagate of observation s + 1, which gathers exactly that state into the
  // thread of slot pick_s, records it on the way; otherwise a one-block launch per observation does.
  const bool fold = path && uses_sums_kernel(pf);
  // Per-observation kernels, enqueued without a host round trip.  With the sums formed inside k_propagate (relative to each
  // observation's reference level), an observation whose max rules its level out puts the series ON HOLD at that
  // observation (err bit 6: every kernel behind it returns at once); the host then redoes that one observation's sums
  // relative to the max (k_tile_sums + k_offspring, the log-weights are in place) and enqueues the rest again.
  size_t s_from = 0;
  while (!pf->last_series) {
    pf->batch_hold = uses_sums_kernel(pf);
    // (small clouds) the resampling of a weighted observation that another weighted observation follows is not launched:
    // the next observation's k_step does it on the way -- one launch per observation instead of two
    const bool merge = !cont && step_eligible(pf) && (!path || fold);   // (k_step indexes ll_t by the observation's index in the filter's series)
    pf->no_fine = merge;   // (k_step consumes the propagate kernel's own unit sums)
    bool deferred = false;
    if (pf->pp_after.size() < T) pf->pp_after.resize(T);
    for (size_t s = s_from; s < T; ++s) {
      const int weighted = pf->h_recs[s].has_obs;
      double* pick_out = (fold && s >= 1) ? pf->d_path + s * (size_t)d : nullptr;
      const uint32_t pick_slot = s >= 1 ? pf->h_recs[s - 1].pick : 0u;
      const bool defer_next = merge && weighted && s + 1 < T && pf->h_recs[s + 1].has_obs;
      pf->h_step_for_resample = base + (uint32_t)s;
      if (deferred) rc = launch_merged_step(pf, pf->d_recs + s, pf->d_ll_t, pf->d_ess_t, pick_out, pick_slot);
      else rc = launch_propagate(pf, pf->d_recs + s, pick_out, pick_slot);
      pf->pp_after[s] = (uint8_t)pf->pp;
      if (!rc && !defer_next) {
        if (weighted) rc = launch_resample(pf, pf->d_recs + s, pf->d_ll_t, pf->d_ess_t, (uint32_t)s);
        else hipLaunchKernelGGL(k_record, dim3(1), dim3(1), 0, pf->stream, pf->sc, pf->d_ll_t, pf->d_ess_t, (uint32_t)s);
      }
      deferred = defer_next;
      if (rc) { pf->batch_hold = false; return rc; }
      if (path && (!fold || s + 1 == T))   // (folded: only the last entry has no following propagate)
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride,
                           (const uint32_t*)(pf->anc_valid ? pf->anc : nullptr), (uint64_t)pf->h_recs[s].pick, d,
                           pf->d_path + (s + 1) * (size_t)d);
    }
    const bool may_hold = pf->batch_hold;
    pf->batch_hold = false;
    if (!may_hold) break;
    Scalars hh;
    HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&hh, pf->sc), hipMemcpyDeviceToHost, pf->stream));
    HIP_TRY(hipStreamSynchronize(pf->stream));
    if (!(hh.err & 64u)) break;
    if (hh.err & 3u) break;            // NaN / unusable weights: reported below
    const size_t sf = (size_t)hh.fail_step - base;              // (the record's index in this call)
    if (hh.fail_step < base || sf >= T) return fail(CSSM_ESTATE, "held series reports observation %u; this call holds %u .. %zu", hh.fail_step, base, base + T - 1);
    hh.err &= ~64u; hh.fail_step = 0xffffffffu;
    HIP_TRY(hipMemcpyAsync(&pf->sc->err, &hh.err, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &hh.fail_step, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
    // host-side state right after the propagate of observation sf (every propagate flips cur, every weighted observation
    // advances wparity)
    pf->cur = (int)(((size_t)cur0 + sf + 1) & 1);
    pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride; pf->anc_valid = false; pf->src2 = nullptr;
    int wp = wpar0;
    for (size_t q = 0; q < sf; ++q) wp = (wp + (pf->h_recs[q].has_obs ? 1 : 0)) % CSSM_MAXSETS;
    pf->wparity = (wp + 1) % CSSM_MAXSETS;   // launch_resample(redo) steps it back to the set the observation's propagate used
    if (sf < pf->pp_after.size() && pf->pp != (int)pf->pp_after[sf]) swap_sets(pf);   // ... and the log-weights / unit sums it wrote
    pf->last_optimistic = true;
    pf->h_step_for_resample = base + (uint32_t)sf;
    rc = launch_resample(pf, pf->d_recs + sf, pf->d_ll_t, pf->d_ess_t, (uint32_t)sf, /*redo=*/true);
    if (rc) return rc;
    if (path && (!fold || sf + 1 == T))
      hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), 0, pf->stream, pf->src, pf->src_stride, (const uint32_t*)pf->anc,
                         (uint64_t)pf->h_recs[sf].pick, d, pf->d_path + (sf + 1) * (size_t)d);
    s_from = sf + 1;
    if (s_from >= T) break;
  }
  HIP_TRY(hipEventRecord(pf->ev1, pf->stream));
  HIP_TRY(hipGetLastError());
  if (!pf->h_sc) HIP_TRY(hipHostMalloc((void**)&pf->h_sc, sizeof(Scalars), hipHostMallocDefault));
  Scalars& h = *pf->h_sc;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  if (ll_t) HIP_TRY(hipMemcpyAsync(ll_t, pf->d_ll_t, T * 8, hipMemcpyDeviceToHost, pf->stream));
  if (ess_t) HIP_TRY(hipMemcpyAsync(ess_t, pf->d_ess_t, T * 4, hipMemcpyDeviceToHost, pf->stream));
  if (path) HIP_TRY(hipMemcpyAsync(path, pf->d_path, (T + 1) * (size_t)d * 8, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  HIP_TRY(hipEventElapsedTime(&pf->last_ms, pf->ev0, pf->ev1));
  prof_collect(pf);
  if (pf->last_series && pf->profile && pf->d_ts) { rc = series_collect_phases(pf, T); if (rc) return rc; }
  pf->t = t[T - 1]; pf->step = base + (uint32_t)T;
  if (h.err & 32u) return fail(CSSM_EHIP, "the series kernel met an ancestor index beyond the cloud (clamped, not dereferenced): internal error");
  if (h.err & 16u) return fail(CSSM_EHIP, "the grid barrier of the series kernel timed out (a block did not arrive); the series was abandoned");
  if ((h.err & 4u) && !(h.err & 1u) && !pf->safe_sums) { *retry = true; return CSSM_OK; }
  if (ll_out) *ll_out = h.ll;
  return check_device_err(pf, h);
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
  if (fn == CSSM_FN_SINCOS2PI || fn == CSSM_FN_FIX) need = 2 * n;
  else if (fn == CSSM_FN_PAIRED_NORMALS) {
    if (n < 5) return fail(CSSM_EINVAL_ARG, "paired normals take {seed, first, step, tag, d}");
    d = (size_t)x[4];
    if (d < 1 || d > CSSM_MAX_DIM) return fail(CSSM_EINVAL_ARG, "d out of range");
    np = n_out / d; need = np * d;
    if (np < 1) return fail(CSSM_EINVAL_ARG, "out is too small");
  } else if (fn < CSSM_FN_EXP || fn > CSSM_FN_FIX) return fail(CSSM_EINVAL_ARG, "unknown contract function %d", fn);
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
  *ms_out = pf->last_ms;
  return CSSM_OK;
}

extern "C" int cssm_pf_set_option(cssm_pf* pf, int option, int value) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (option == CSSM_OPT_EXACT_OFFSPRING) { pf->opt_exact = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_FUSED_SUMS) { pf->opt_fused = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_SERIES_KERNEL) { pf->opt_series = value ? 1 : 0; return CSSM_OK; }
  if (option == CSSM_OPT_ONE_LAUNCH) { pf->opt_step = value < 0 ? -1 : (value ? 1 : 0); return CSSM_OK; }
  if (option == CSSM_OPT_WHOLE_TILES) {   // launch geometry only: the arrays hold up to four sub-units per unit either way
    if (pf->sharded) return fail(CSSM_ESTATE, "sharded handles always run whole tiles");
    pf->opt_whole = value < 0 ? 0 : (value > 3 ? 3 : value);
    pf->split = value ? 1u : auto_split(pf);
    return CSSM_OK;
  }
  if (option == CSSM_OPT_RESAMPLER) {
    if (value < CSSM_RESAMPLE_SYSTEMATIC || value > CSSM_RESAMPLE_MULTINOMIAL) return fail(CSSM_EINVAL_ARG, "unknown resampler %d", value);
    if (pf->sharded && value != CSSM_RESAMPLE_SYSTEMATIC) return fail(CSSM_ESTATE, "sharded handles resample systematically");
    pf->resampler = value;
    return CSSM_OK;
  }
  return fail(CSSM_EINVAL_ARG, "unknown option %d", option);
}

extern "C" int cssm_pf_profile(cssm_pf* pf, int enable) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  pf->profile = enable != 0;
  pf->profile_level = enable;
  pf->prof_used = 0;
  for (int k = 0; k < CSSM_NKERNELS; ++k) { pf->prof_ms[k] = 0.0; pf->prof_cnt[k] = 0; }
  return CSSM_OK;
}

extern "C" int cssm_pf_profile_read(cssm_pf* pf, double* total_ms, uint64_t* launches) {
  if (!pf || !total_ms || !launches) return fail(CSSM_EINVAL_ARG, "null argument");
  for (int k = 0; k < CSSM_NKERNELS; ++k) { total_ms[k] = pf->prof_ms[k]; launches[k] = pf->prof_cnt[k]; }
  return CSSM_OK;
}

extern "C" int cssm_pf_series_phases(cssm_pf* pf, int* used_series, double* phase_us, uint64_t* weighted_steps) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (used_series) *used_series = pf->last_series ? 1 : 0;
  if (phase_us) for (int k = 0; k < 4; ++k) phase_us[k] = pf->ser_phase_us[k];
  if (weighted_steps) *weighted_steps = pf->ser_phase_steps;
  return CSSM_OK;
}

extern "C" int cssm_pf_series_stamps(cssm_pf* pf, uint64_t* out, size_t cap, uint32_t* blocks, uint32_t* steps) {
  if (!pf) return fail(CSSM_EINVAL_ARG, "null handle");
  if (blocks) *blocks = pf->ts_blocks;
  if (steps) *steps = (uint32_t)pf->ts_T;
  const size_t n = (size_t)pf->ts_blocks * pf->ts_T * CSSM_SER_TS_PER_STEP;
  if (out) {
    if (!pf->d_ts || !pf->last_series || cap < n) return fail(CSSM_ESTATE, "no stamps of a profiled series run (or the buffer is too small: %zu words)", n);
    HIP_TRY(hipMemcpy(out, pf->d_ts, n * 8, hipMemcpyDeviceToHost));
  }
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

extern "C" int cssm_pf_get_logw(cssm_pf* pf, double* out) {
  if (!pf || !out) return fail(CSSM_EINVAL_ARG, "null argument");
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
  build_rec(pf, time, time, 0.0, 0, pf->step, &hrec);   // F(t) of the requested time for f(x, t)
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
  if (eta_of_mean) {   // meanEta = link(f(stateMean, t)), :420
    double g = 0.0, acc = 0.0;
    for (int k = 0; k < d; ++k) {
      const int fm = pf->mk.fmode(k);
      if (fm == FM_START) acc = hrec.fco[k] * hout[k]; else if (fm == FM_ADD) acc = acc + hrec.fco[k] * hout[k];
      if (pf->mk.leaf_end(k)) g = pf->mk.first_leaf(k) ? acc : g + acc;
    }
    *eta_of_mean = host_link(pf->obs_kind, g);
  }
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
  int rc = ensure_recs(pf, T);
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
  for (size_t s = 0; s < T; ++s) { build_rec(pf, tp, t[s], y[s], has ? has[s] : 1, (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = build_fsub(pf, 0, T, true);
  if (rc) return rc;
  rc = (hipMemcpyAsync(pf->d_recs, pf->h_recs, T * sizeof(StepRec), hipMemcpyHostToDevice, pf->stream) == hipSuccess) ? CSSM_OK : fail(CSSM_EHIP, "record upload");
  Scalars h;
  for (int attempt = 0; attempt < 2 && !rc; ++attempt) {   // second attempt: see run_filter
    pf->safe_sums = (attempt == 1);
    pf->state[0] = hx;                                   // X1_0 = the initial cloud
    rc = launch_init(pf, t0);
    for (size_t s = 0; s < T && !rc; ++s) {
      // step s+1 reads X1_s through anc_s (launch_propagate uses pf->anc / pf->anc_valid) and writes X1_{s+1}
      pf->state[pf->cur ^ 1] = hx + (s + 1) * slab;
      pf->anc = hanc + s * pf->stride;
      const int w = pf->h_recs[s].has_obs;
      pf->h_step_for_resample = (uint32_t)s;
      rc = launch_propagate(pf, pf->d_recs + s);
      if (!rc && w) {
        pf->anc = hanc + (s + 1) * pf->stride;
        rc = launch_resample(pf, pf->d_recs + s);
        weighted[s + 1] = 1;
      }
    }
    if (!rc && hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream) != hipSuccess) rc = fail(CSSM_EHIP, "scalars");
    if (!rc && hipStreamSynchronize(pf->stream) != hipSuccess) rc = fail(CSSM_EHIP, "forward pass");
    if (rc || !(h.err & 4u) || (h.err & 1u)) break;
  }
  pf->safe_sums = false;
  if (!rc) rc = check_device_err(pf, h);
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
#define RS_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(CSSM_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); goto done; } } while (0)
  RS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  RS_TRY(hipMalloc(&d_w, stride * 8)); RS_TRY(hipMalloc(&d_end, stride * 4)); RS_TRY(hipMalloc(&d_anc, stride * 4));
  RS_TRY(hipMalloc(&tS, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tS2, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tP, ntiles * sizeof(cssm_u128)));
  RS_TRY(hipMalloc(&sc, sizeof(Scalars))); RS_TRY(hipMalloc(&d_rec, sizeof(StepRec))); RS_TRY(hipMalloc(&d_tab, sizeof(CSSM_TAB)));
  if (kind == CSSM_RESAMPLE_MULTINOMIAL) RS_TRY(hipMalloc(&d_cum, stride * 8));
  RS_TRY(hipMemcpyAsync(d_tab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice, st));
  RS_TRY(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
  RS_TRY(hipMemcpyAsync(d_w, w, n * 8, hipMemcpyHostToDevice, st));
  RS_TRY(hipMemcpyAsync(d_rec, &hrec, sizeof hrec, hipMemcpyHostToDevice, st));
  {
    const int tgrid = (int)nunits;
    hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, st, d_w, (uint64_t)n, sc, tS, tS2, ntiles, sup, nunits, 1, -1, (const double*)nullptr, d_tab,
                       (const StepRec*)d_rec);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tS, tS2, tP, nunits, sc, (uint64_t)n, 1, (double*)nullptr, (int32_t*)nullptr, 0u,
                       (const double*)nullptr, (unsigned long long*)nullptr, 0);
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

// ------------------------------------------------------------------------------------ sharded stages
// One process per GPU; the collectives between the stages belong to the caller (RCCL through
// torch.distributed).  See include/cssm_pf.h for the sequence.

// level of the step from the all-gathered order keys (word 4 of every rank's 5 words)
__global__ void k_import_level(Scalars* sc, const unsigned long long* __restrict__ all5, int world, const StepRec* __restrict__ rec) {
  unsigned long long key = 0ull;
  for (int r = 0; r < world; ++r) { const unsigned long long k = all5[5 * r + 4]; key = (k > key) ? k : key; }
  sc->gmax = cssm_order_unkey(key);
  sc->ref = cssm_ref_choose(rec->ref, sc->gmax);   // the level every kernel of the step agrees on (k_tile_sums applies the same rule)
}

// For every destination rank q (owner of slots [q*n_per, min((q+1)*n_per, N))): the contiguous
// range of LOCAL particles that own at least one of q's slots.  One thread per q.
__global__ void k_send_ranges(const uint32_t* __restrict__ endslot, uint64_t n_local, const Scalars* __restrict__ sc,
                              const StepRec* __restrict__ rec, uint64_t n_global, int rank, int world, uint64_t n_per,
                              long long* __restrict__ first, long long* __restrict__ count) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= world) return;
  uint64_t b_lo = (uint64_t)q * n_per, b_hi = b_lo + n_per;
  if (b_lo > n_global) b_lo = n_global;
  if (b_hi > n_global) b_hi = n_global;
  uint64_t e_before = 0;   // end slot of the last particle of the previous rank
  if (rank > 0) e_before = cssm_sys_count(cssm_u128_to_double(sc->S_off) / cssm_u128_to_double(sc->S_tot), rec->u, n_global);
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
// record of the step propagated last: a ring of 64 for streaming steps, the whole series after shard_begin
static size_t last_rec_slot(const cssm_pf* pf) { return pf->series ? (size_t)(pf->step - 1) : (size_t)((pf->step - 1) % 64); }

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
  return launch_init(pf, t0);
}

extern "C" int cssm_pf_shard_propagate(cssm_pf* pf, double t, double obs, int has_obs, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!pf->initialised) return fail(CSSM_ESTATE, "shard_propagate before shard_init");
  // one record slot per step, round-robin, so that an in-flight step never sees its record overwritten
  rc = ensure_recs(pf, 64);
  if (rc) return rc;
  const size_t slot = pf->step % 64;
  build_rec(pf, pf->t, t, obs, has_obs, pf->step, &pf->h_recs[slot]);
  rc = build_fsub(pf, slot, 1, true);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pf->d_recs + slot, pf->h_recs + slot, sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  if (pf->series) return fail(CSSM_ESTATE, "a series begun with shard_begin is stepped with shard_propagate_at");
  rc = shard_prepare_step(pf, pf->d_recs + slot, pf->h_recs[slot].has_obs, sums5_dev);
  if (rc) return rc;
  pf->t = t;
  pf->step++;
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_sums(cssm_pf* pf, const uint64_t* all_sums5_dev, int world, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!all_sums5_dev || !sums5_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  if (world < 1 || world > 64) return fail(CSSM_ESHARD, "world %d", world);
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits;
  hipLaunchKernelGGL(k_import_level, dim3(1), dim3(1), 0, pf->stream, pf->sc, (const unsigned long long*)all_sums5_dev, world,
                     (const StepRec*)(pf->d_recs + slot));
  hipLaunchKernelGGL(k_tile_sums, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc, pf->tileS, pf->tileS2, pf->ntiles,
                     pf->sup, pf->nunits, 0, -1, (const double*)nullptr, pf->d_logtab, (const StepRec*)(pf->d_recs + slot));
  // word 4 (the max key) of sums5_dev is left as shard_propagate wrote it
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, pf->stream, pf->tileS, pf->tileS2, pf->tileP, pf->nunits, pf->sc, pf->n_global, 0,
                     (double*)nullptr, (int32_t*)nullptr, 0u, (const double*)nullptr, (unsigned long long*)sums5_dev, 0);
  HIP_TRY(hipGetLastError());
  pf->last_optimistic = false;
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
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits;
  const int optimistic = pf->last_optimistic ? 1 : 0;
  // the rank's own particles write their runs inside the rank's slots straight into anc (indexed from the first
  // own slot); the end slots are kept for the send ranges; slots owned by other ranks' particles are filled by adopt
  hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_SYSTEMATIC>), dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc,
                     (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, pf->d_recs + slot, pf->n_global, pf->endslot,
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, 0, 0, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,
                     (const unsigned long long*)all_sums5_dev, rank, world, optimistic ? (int)pf->split : 1, pf->seed, (double*)nullptr, pf->d_logtab,
                     optimistic, (unsigned long long*)redo_flag_dev, (uint32_t)pf->first, (uint32_t)(pf->first + pf->n));
  pf->send_first_dev = (const long long*)send_first_dev; pf->send_count_dev = (const long long*)send_count_dev;
  hipLaunchKernelGGL(k_send_ranges, dim3(1), dim3(64), 0, pf->stream, pf->endslot, pf->n, pf->sc, pf->d_recs + slot, pf->n_global, rank, world,
                     n_per, (long long*)send_first_dev, (long long*)send_count_dev);
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
  int rc = launch_propagate(pf, d_rec);
  if (rc) return rc;
  if (weighted && sums5_dev) {   // (sums5_dev == nullptr: the single-collective exchange totals the sums in k_boundary_pack)
    // the rank's totals of the sub-unit sums k_propagate formed and the order key of its max -> 5 words for the all-gather
    const uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / pf->split;
    const uint32_t nsub = (uint32_t)((pf->n + chunk - 1) / chunk);
    if (!pf->last_optimistic) {   // LGCP: only the max travels
      HIP_TRY(hipMemsetAsync(pf->tileS, 0, (size_t)nsub * sizeof(cssm_u128), pf->stream));
      HIP_TRY(hipMemsetAsync(pf->tileS2, 0, (size_t)nsub * sizeof(cssm_u128), pf->stream));
    }
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, pf->stream, pf->tileS, pf->tileS2, pf->tileP, nsub, pf->sc, pf->n_global, 0,
                       (double*)nullptr, (int32_t*)nullptr, 0u, (const double*)nullptr, (unsigned long long*)sums5_dev, 1);
    HIP_TRY(hipGetLastError());
  }
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_begin(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!t || !y) return fail(CSSM_EINVAL_ARG, "null argument");
  if (T < 1) return fail(CSSM_EINVAL_ARG, "empty data");
  rc = ensure_recs(pf, T);
  if (rc) return rc;
  if (pf->need_cap < T) {
    if (pf->d_need) (void)hipFree(pf->d_need);
    pf->d_need = nullptr;
    HIP_TRY(hipMalloc(&pf->d_need, T * 4));
    pf->need_cap = T;
  }
  HIP_TRY(hipMemsetAsync(pf->d_need, 0, T * 4, pf->stream));
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];
  double tp = t0;
  for (size_t s = 0; s < T; ++s) { build_rec(pf, tp, t[s], y[s], has_obs ? has_obs[s] : 1, (uint32_t)s, &pf->h_recs[s]); tp = t[s]; }
  rc = build_fsub(pf, 0, T, true);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pf->d_recs, pf->h_recs, T * sizeof(StepRec), hipMemcpyHostToDevice, pf->stream));
  rc = launch_init(pf, t0);
  if (rc) return rc;
  pf->series = true;
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_propagate_at(cssm_pf* pf, size_t s, uint64_t* sums5_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!pf->series || !pf->initialised) return fail(CSSM_ESTATE, "shard_propagate_at before shard_begin");
  if (s >= pf->h_recs_cap || s != pf->step) return fail(CSSM_ESTATE, "steps of a series run in order (expected %u)", pf->step);
  rc = shard_prepare_step(pf, pf->d_recs + s, pf->h_recs[s].has_obs, sums5_dev);
  if (rc) return rc;
  pf->step++;
  if (pf->snaps.size() <= s) pf->snaps.resize(s + 1);
  pf->snaps[s] = {pf->cur, pf->src, pf->src_stride, pf->src2, pf->src2_stride, pf->n_split, pf->anc_valid, pf->last_optimistic, pf->step, pf->t};
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
  if (!(h.err & 8u) || h.fail_step == 0xffffffffu || h.fail_step >= pf->snaps.size())
    return fail(CSSM_ESTATE, "no resumable capacity miss is recorded");
  if (h.err & 7u) return fail(CSSM_ESTATE, "the series has other errors (bits %u)", h.err);
  const uint32_t s = h.fail_step;
  h.err &= ~8u; h.fail_step = 0xffffffffu;
  // only err and fail_step change on the device (ll, ess, sums stay what the last completed observation left)
  HIP_TRY(hipMemcpyAsync(&pf->sc->err, &h.err, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipMemcpyAsync(&pf->sc->fail_step, &h.fail_step, sizeof(uint32_t), hipMemcpyHostToDevice, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  const cssm_pf::Snap& q = pf->snaps[s];
  pf->cur = q.cur; pf->src = q.src; pf->src_stride = q.src_stride; pf->src2 = q.src2; pf->src2_stride = q.src2_stride;
  pf->n_split = q.n_split; pf->anc_valid = q.anc_valid; pf->last_optimistic = q.last_optimistic; pf->step = q.step; pf->t = q.t;
  *fail_step_out = s;
  return CSSM_OK;
}

// ---- single-collective exchange (k_boundary_pack / k_expand_spec in cssm_kernels.hip.h)

extern "C" int64_t cssm_pf_shard_spec_segment(const cssm_pf* pf, int64_t cap) {
  return (pf && cap >= 1) ? (int64_t)spec_seg(pf->d, (long long)cap) : 0;
}

extern "C" int cssm_pf_shard_boundary_pack(cssm_pf* pf, int rank, int world, int64_t cap, double* send_buf_dev) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!send_buf_dev) return fail(CSSM_EINVAL_ARG, "null argument");
  if (cap < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d / cap %lld", rank, world, (long long)cap);
  // !last_optimistic: the sums were formed by cssm_pf_shard_sums relative to the level chosen with the all-gathered max
  const size_t slot = last_rec_slot(pf);
  const uint64_t chunk = (uint64_t)pf->sup * CSSM_TILE / pf->split;
  const uint32_t nsub = (uint32_t)((pf->n + chunk - 1) / chunk);
  const long long cnt = std::min<long long>((long long)pf->n, (long long)cap);
  const int tiles = (int)((cnt + CSSM_TILE - 1) / CSSM_TILE);
  hipLaunchKernelGGL(k_boundary_pack, dim3(tiles + 1, world), dim3(CSSM_BLOCK), 0, pf->stream, pf->state[pf->cur], pf->stride, pf->logw, pf->n, pf->d,
                     world, rank, (long long)cap, pf->d_recs + slot, (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, nsub,
                     (const Scalars*)pf->sc, send_buf_dev, chunk, pf->last_optimistic ? 0 : 1);
  HIP_TRY(hipGetLastError());
  return CSSM_OK;
}

extern "C" int cssm_pf_shard_adopt_spec(cssm_pf* pf, const double* recv_buf_dev, int rank, int world, int64_t cap) {
  int rc = shard_check(pf);
  if (rc) return rc;
  if (!recv_buf_dev) return fail(CSSM_EINVAL_ARG, "recv_buf_dev is null");
  if (cap < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(CSSM_ESHARD, "rank %d / world %d / cap %lld", rank, world, (long long)cap);
  const uint64_t n_per = (pf->n_global + (uint64_t)world - 1) / (uint64_t)world;
  if (pf->first != (uint64_t)rank * n_per) return fail(CSSM_ESHARD, "rank %d must own particles from %llu", rank, (unsigned long long)((uint64_t)rank * n_per));
  const size_t slot = last_rec_slot(pf);
  const int tgrid = (int)pf->nunits;
  const long long seg = spec_seg(pf->d, (long long)cap);
  // the 5 words of every rank are the header words 1..5 of its segment (the all-to-all delivered this rank's own too)
  const unsigned long long* all5 = reinterpret_cast<const unsigned long long*>(recv_buf_dev) + 1;
  const uint32_t n_split = (uint32_t)pf->n;
  hipLaunchKernelGGL(k_offspring_expand_spec, dim3(tgrid), dim3(CSSM_BLOCK), 0, pf->stream, pf->logw, pf->n, pf->sc,
                     (const cssm_u128*)pf->tileS, (const cssm_u128*)pf->tileS2, (const StepRec*)(pf->d_recs + slot), pf->n_global, pf->endslot,
                     pf->anc, pf->ntiles, pf->sup, pf->nunits, 0, 0, (double*)nullptr, (int32_t*)nullptr, 0u, pf->opt_exact,
                     all5, rank, world, (int)pf->split, pf->seed, (double*)nullptr, (const double*)pf->d_logtab,
                     pf->last_optimistic ? 2 : 0, (unsigned long long*)(pf->d_xch + 128), (uint32_t)pf->first, (uint32_t)(pf->first + pf->n), (uint32_t)seg,
                     recv_buf_dev, (long long)cap, pf->d, n_split);
  HIP_TRY(hipGetLastError());
  pf->src = pf->state[pf->cur]; pf->src_stride = pf->stride;
  pf->src2 = recv_buf_dev; pf->src2_stride = 0; pf->n_split = n_split; pf->anc_valid = true;   // stride 0 = rows of d + 1
  return CSSM_OK;
}

// ll, ess and the sticky bits of a series run with the single-collective exchange: bit 2 (value 4) = some observation's
// reference level was ruled out by the max, bit 3 (value 8) = a capacity miss that was not resumed.  Either bit means the
// numbers are not the filter's (ShardedFilter moves on to its next plan); `need` (optional, T entries): diagnostics.
extern "C" int cssm_pf_shard_status(cssm_pf* pf, double* ll_out, int32_t* ess_out, uint32_t* bits_out, uint32_t* need, size_t T) {
  int rc = shard_check(pf);
  if (rc) return rc;
  rc = bounded_sync(pf);   // (first: a copy into pageable memory would wait for the stream without a bound)
  if (rc) return rc;
  Scalars h;
  HIP_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&h, pf->sc), hipMemcpyDeviceToHost, pf->stream));
  if (need && pf->d_need && T <= pf->need_cap) HIP_TRY(hipMemcpyAsync(need, pf->d_need, T * 4, hipMemcpyDeviceToHost, pf->stream));
  HIP_TRY(hipStreamSynchronize(pf->stream));
  if (ll_out) *ll_out = h.ll;
  if (ess_out) *ess_out = h.ess;
  if (bits_out) *bits_out = h.err & 12u;
  h.err &= ~12u;
  return check_device_err(pf, h);
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
  // the copy already mapped into this process (the host's framework usually brings one: two RCCL instances side by side
  // would each keep their own topology and IPC state), else the ROCm installation's
  if (FILE* maps = fopen("/proc/self/maps", "r")) {
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
    for (size_t s = s_begin; s < s_end; ++s) {
      rc = cssm_pf_shard_propagate_at(pf, s, level_from_max ? sums5_dev : nullptr);
      if (rc) return rc;
      if (!weighted[s]) continue;
      if (level_from_max) {
        const int rg = a->AllGather(sums5_dev, all_sums5_dev, 5, kNcclUint64, comm, pf->stream);
        if (rg) { if (a->CommAbort) (void)a->CommAbort(comm); pf->last_comm = nullptr; return rccl_fail(a, "ncclAllGather", rg); }
        rc = cssm_pf_shard_sums(pf, all_sums5_dev, world, sums5_dev);
        if (rc) return rc;
      }
      rc = cssm_pf_shard_boundary_pack(pf, rank, world, cap, send_buf_dev);
      if (rc) return rc;
      const int r = trimmed ? a->AllToAllv(send_buf_dev, counts.data(), displs.data(), recv_buf_dev, counts.data(), displs.data(), kNcclFloat64,
                                           comm, pf->stream)
                            : a->AllToAll(send_buf_dev, recv_buf_dev, sseg, kNcclFloat64, comm, pf->stream);
      if (r) {   // this rank will not enqueue the rest: end the collectives its peers may already wait in
        if (a->CommAbort) (void)a->CommAbort(comm);
        pf->last_comm = nullptr;
        return rccl_fail(a, trimmed ? "ncclAllToAllv" : "ncclAllToAll", r);
      }
      rc = cssm_pf_shard_adopt_spec(pf, recv_buf_dev, rank, world, cap);
      if (rc) return rc;
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
  return check_device_err(pf, h);
}

// ------------------------------------------------------------------------------------ PMMH host loop

struct OwnedDesc {
  std::vector<cssm_leaf_desc> leaves;
  std::vector<std::vector<double>> store;
  cssm_model_desc desc;
  std::vector<double*> slots;   // Parameters.flattenParams order, model/Parameters.scala:88-95
};

static int own_desc(const cssm_model_desc* in, OwnedDesc* o) {
  if (!in || !in->leaves || in->n_leaves < 1 || in->n_leaves > CSSM_MAX_LEAVES) return fail(CSSM_EINVAL_DESC, "bad descriptor");
  o->leaves.assign(in->leaves, in->leaves + in->n_leaves);
  o->store.clear();
  o->store.reserve((size_t)in->n_leaves * 5);
  for (auto& L : o->leaves) {
    auto take = [&](const double*& p, int n) {
      o->store.emplace_back(p && n > 0 ? std::vector<double>(p, p + n) : std::vector<double>());
      p = o->store.back().empty() ? nullptr : o->store.back().data();
    };
    take(L.m0, L.n_m0); take(L.c0, L.n_c0); take(L.mu, L.n_mu); take(L.phi, L.n_phi); take(L.sigma, L.n_sigma);
  }
  o->desc = *in;
  o->desc.leaves = o->leaves.data();
  o->slots.clear();
  for (auto& L : o->leaves) {
    auto push = [&](const double* p, int n) { for (int i = 0; i < n; ++i) o->slots.push_back(const_cast<double*>(p) + i); };
    if (L.has_scale) o->slots.push_back(&L.scale);
    push(L.m0, L.n_m0); push(L.c0, L.n_c0);
    if (L.sde_kind == CSSM_SDE_GEN_BROWNIAN) push(L.mu, L.n_mu);                                   // m0 ++ c0 ++ mu ++ sigma
    else if (L.sde_kind == CSSM_SDE_OU || L.sde_kind == CSSM_SDE_EULER_AFFINE) { push(L.phi, L.n_phi); push(L.mu, L.n_mu); }  // m0 ++ c0 ++ phi ++ mu ++ sigma
    push(L.sigma, L.n_sigma);
  }
  return CSSM_OK;
}

extern "C" int cssm_desc_flatten(const cssm_model_desc* desc, double* theta, size_t cap, size_t* n_theta) {
  OwnedDesc o;
  int rc = own_desc(desc, &o);
  if (rc) return rc;
  if (n_theta) *n_theta = o.slots.size();
  if (theta) for (size_t i = 0; i < o.slots.size() && i < cap; ++i) theta[i] = *o.slots[i];
  return CSSM_OK;
}

// mhStep, model/PMMH.scala:68-81; init ll = -1e99 (:121); proposal Parameters.perturb(delta),
// model/Parameters.scala:65-67; the current ll is reused, never re-estimated (:63-66).
extern "C" int cssm_pmmh_run(cssm_pf* pf, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta,
                             const double* t, const double* y, const uint8_t* has_obs, size_t T, uint64_t seed,
                             size_t n_iters, double* ll, double* theta, int32_t* accepted, double* last_state) {
  if (!pf || !theta0 || !ll || !theta || !accepted || !last_state) return fail(CSSM_EINVAL_ARG, "null argument");
  OwnedDesc o;
  int rc = own_desc(desc, &o);
  if (rc) return rc;
  if (o.slots.size() != n_theta) return fail(CSSM_EINVAL_ARG, "theta0 has %zu entries, the descriptor flattens to %zu", n_theta, o.slots.size());
  const int d = pf->d;
  std::vector<double> cur(theta0, theta0 + n_theta), prop(n_theta), path((T + 1) * (size_t)d), cur_state(d, 0.0);
  double cur_ll = -1e99;
  int32_t acc = 0;
  const double sd = std::sqrt(delta);
  for (size_t it = 0; it < n_iters; ++it) {
    for (size_t j = 0; j < n_theta; j += 2) {                  // propParams <- proposal(s.params)
      double z0, z1;
      cssm_normal_pair(cssm_philox_draw(seed, it, (uint32_t)(j / 2), CSSM_STREAM_HOST, 0), CSSM_LOG_TAB, &z0, &z1);
      prop[j] = cur[j] + sd * z0;
      if (j + 1 < n_theta) prop[j + 1] = cur[j + 1] + sd * z1;
    }
    for (size_t j = 0; j < n_theta; ++j) *o.slots[j] = prop[j];
    rc = cssm_pf_set_params(pf, &o.desc);
    if (rc) return rc;
    cssm_pf_reseed(pf, cssm_derive_key(seed, (uint64_t)it + 1));   // (never seed + it: include/cssm_numerics.h)
    double pll = 0.0;
    rc = run_filter(pf, t, y, has_obs, T, &pll, nullptr, nullptr, path.data());   // state = pf(propParams)
    if (rc == CSSM_ENONFINITE) pll = -cssm_inf();              // a proposal the filter cannot weigh is rejected
    else if (rc) return rc;
    const double a = pll - cur_ll;                             // logTransition = prior = 0
    const cssm_u32x4 b = cssm_philox_draw(seed, it, 0xffffffffu, CSSM_STREAM_HOST, 1);
    const double uu = cssm_u01_open0(b.v[0], b.v[1]);
    if (cssm_log(uu) < a) {                                    // :75
      cur_ll = pll; cur = prop; ++acc;
      memcpy(cur_state.data(), path.data() + T * (size_t)d, d * 8);
    }
    ll[it] = cur_ll; accepted[it] = acc;
    memcpy(theta + it * n_theta, cur.data(), n_theta * 8);
    memcpy(last_state + it * (size_t)d, cur_state.data(), d * 8);
  }
  return CSSM_OK;
}

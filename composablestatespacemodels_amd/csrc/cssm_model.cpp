// cssm_model.cpp -- the host-only part of libcssm_pf: error strings, descriptor validation and constraint transforms, the
// per-observation records (build_rec), Parameters.flattenParams order, and the PMMH host loop over the public C ABI.  No HIP
// here: the file compiles with any C++17 compiler, which is how its sanitizer build works (oracle/Makefile, target `san`,
// with the three ABI calls the PMMH loop makes stubbed by the test harness).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cssm_records.h"

// ------------------------------------------------------------------------------------ errors

static thread_local std::string g_err;

int cssm_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define fail cssm_fail

extern "C" const char* cssm_last_error(void) { return g_err.c_str(); }
extern "C" const char* cssm_version(void) { return "cssm_pf 0.5 (gfx950, numerics contract v8)"; }

// ------------------------------------------------------------------------------------ model

static double rep(const double* v, int n, int i) { return v[i % n]; }               // Sde.buildParamRepeat, model/Sde.scala:177-179
static double logistic(double x) { return 1.0 / (1.0 + cssm_exp(-x)); }              // model/SdeParameters.scala:214-216

// Constraint transforms of the SDE constructors: model/Sde.scala:70-73 (GenBrownian),
// :99-102 (Brownian), :133-137 (OU, logistic applied to the stored value).
static int build_model_into(HostModel* pf, const cssm_model_desc* desc);

// Validate and translate the descriptor into a scratch handle first; the real handle changes only when everything
// checked out (a failing cssm_pf_set_params leaves the previous parameters in force, whole).  `update`: the call
// re-parameterises an existing handle -- the STRUCTURE (leaves, dimensions, SDE kinds, f kinds, periods, observation
// model, LGCP precision, Student-t df) must be the one the handle was created with: kernels, buffers and the sharding of
// work were sized for it.
int cssm_build_model(HostModel* pf, const cssm_model_desc* desc, bool update) {
  HostModel tmp;
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

static int build_model_into(HostModel* pf, const cssm_model_desc* desc) {
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

// Everything of one observation that does not depend on the particle: transition coefficients
// (model/Sde.scala:88-91,117-119,139-146), F(t) (model/Model.scala:217-223), the observation
// constants, the resampling uniform (model/Resampling.scala:66) and the sampleOne index (:152).
void cssm_build_rec(const HostModel* pf, double t_prev, double t, double y, int has_obs, uint32_t step, StepRec* r) {
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
  // (LGCP: predicted on the device from the previous weighted observation's max -- StepRec::predict; NaN until it is)
  r->ref = (pf->obs_kind == CSSM_OBS_LGCP) ? cssm_nan() : cssm_ref_level(pf->obs_kind, y, pf->scale_sd, (double)pf->obs_df);
  r->predict = (pf->obs_kind == CSSM_OBS_LGCP) ? 1u : 0u;
  const cssm_u32x4 bu = cssm_philox_draw(pf->seed, 0, step, CSSM_STREAM_U, 0);
  r->u = cssm_u01(bu.v[0], bu.v[1]);
  const int32_t pr = (int32_t)cssm_philox_draw(pf->seed, 0, step + 1, CSSM_STREAM_PICK, 0).v[0];
  const uint32_t pa = pr < 0 ? (uint32_t)0 - (uint32_t)pr : (uint32_t)pr;
  r->pick = (uint32_t)((uint64_t)pa % pf->n_global);
}

// FilterLgcp.calcWeight evaluates f at EVERY simulated time tau_s = t + s delta, the clock starting at the observation's
// time (model/ParticleFilter.scala:193-205, :215; model/Sde.scala:57-66 -- a reference quirk that only shows when f depends
// on time, i.e. with a seasonal leaf).  For such models the coefficients c_k(tau_s) of records [first, first + count) are
// tabulated here (they depend on (t, s) only; tau is accumulated by repeated addition exactly as the oracle does); the
// caller uploads the table behind the records, the kernel reads row s of its observation.
int cssm_build_fsub_table(const HostModel* pf, StepRec* recs, size_t first, size_t count, std::vector<double>& table) {
  if (!pf->lgcp_tdep) return CSSM_OK;
  const int d = pf->d;
  for (size_t q = first; q < first + count; ++q) {
    StepRec* r = &recs[q];
    r->fsub_off = 0;
    if (r->n_sub <= 0) continue;
    if (table.size() + (size_t)r->n_sub * d > ((size_t)1 << 27))
      return fail(CSSM_ENOMEM, "the sub-step table of a time-dependent LGCP model would exceed 1 GiB (%d sub-steps at observation %zu)", r->n_sub, q);
    r->fsub_off = (uint32_t)table.size();
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
        table.push_back(v);
      }
    }
  }
  return CSSM_OK;
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

extern "C" int cssm_model_structure(const cssm_model_desc* desc, uint32_t* words_out, int32_t* d_out) {
  if (!desc || !words_out) return fail(CSSM_EINVAL_ARG, "null argument");
  HostModel tmp;
  int rc = build_model_into(&tmp, desc);
  if (rc) return rc;
  for (int i = 0; i < CSSM_MAX_DIM / 4; ++i) words_out[i] = tmp.mk.comp[i];
  if (d_out) *d_out = tmp.d;
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
// One chain's state, so that the sequential driver (cssm_pmmh_run) and the batched one (cssm_pmmh_run_batched: B chains whose
// filters of an iteration run as one batch, cssm_batch.hip) make the same proposals, key their filters alike and decide alike.
struct cssm_pmmh_chain {
  OwnedDesc o;
  std::vector<double> cur, prop, cur_state;
  double cur_ll = -1e99, sd = 0.0;
  int32_t acc = 0;
  uint64_t seed = 0;
  size_t n_theta = 0;
  int d = 0;
};
int cssm_pmmh_chain_create(const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, uint64_t seed, int d, cssm_pmmh_chain** out) {
  cssm_pmmh_chain* c = new cssm_pmmh_chain();
  int rc = own_desc(desc, &c->o);
  if (!rc && c->o.slots.size() != n_theta) rc = fail(CSSM_EINVAL_ARG, "theta0 has %zu entries, the descriptor flattens to %zu", n_theta, c->o.slots.size());
  if (rc) { delete c; return rc; }
  c->cur.assign(theta0, theta0 + n_theta); c->prop.resize(n_theta); c->cur_state.assign((size_t)d, 0.0);
  c->sd = std::sqrt(delta); c->seed = seed; c->n_theta = n_theta; c->d = d;
  *out = c;
  return CSSM_OK;
}
void cssm_pmmh_chain_destroy(cssm_pmmh_chain* c) { delete c; }
void cssm_pmmh_chain_set_current(cssm_pmmh_chain* c, const double* theta) { c->cur.assign(theta, theta + c->n_theta); }
void cssm_pmmh_chain_set_proposal(cssm_pmmh_chain* c, const double* theta) { c->prop.assign(theta, theta + c->n_theta); }
const double* cssm_pmmh_chain_current(const cssm_pmmh_chain* c) { return c->cur.data(); }
const double* cssm_pmmh_chain_proposal(const cssm_pmmh_chain* c) { return c->prop.data(); }
int32_t cssm_pmmh_chain_accepted(const cssm_pmmh_chain* c) { return c->acc; }
// propParams <- proposal(s.params) of iteration `it`: the descriptor to filter under, and the filter's Philox key
const cssm_model_desc* cssm_pmmh_chain_propose(cssm_pmmh_chain* c, size_t it, uint64_t* key_out) {
  for (size_t j = 0; j < c->n_theta; j += 2) {
    double z0, z1;
    cssm_normal_pair_of(c->seed, it, (uint32_t)(j / 2), CSSM_STREAM_HOST, 0, CSSM_LOG_TAB, &z0, &z1);
    c->prop[j] = c->cur[j] + c->sd * z0;
    if (j + 1 < c->n_theta) c->prop[j + 1] = c->cur[j + 1] + c->sd * z1;
  }
  for (size_t j = 0; j < c->n_theta; ++j) *c->o.slots[j] = c->prop[j];
  *key_out = cssm_derive_key(c->seed, (uint64_t)it + 1);      // (never seed + it: include/cssm_numerics.h)
  return &c->o.desc;
}
// accept iff log(u) < ll' - ll (:75; logTransition = prior = 0); the iteration's output row
void cssm_pmmh_chain_decide(cssm_pmmh_chain* c, size_t it, double pll, const double* last_path_row, double* ll_out, double* theta_out, int32_t* acc_out,
                            double* state_out) {
  const double a = pll - c->cur_ll;
  const cssm_u32x4 b = cssm_philox_draw(c->seed, it, 0xffffffffu, CSSM_STREAM_HOST, 1);
  const double uu = cssm_u01_open0(b.v[0], b.v[1]);
  if (cssm_log(uu) < a) {
    c->cur_ll = pll; c->cur = c->prop; ++c->acc;
    memcpy(c->cur_state.data(), last_path_row, (size_t)c->d * 8);
  }
  *ll_out = c->cur_ll; *acc_out = c->acc;
  memcpy(theta_out, c->cur.data(), c->n_theta * 8);
  memcpy(state_out, c->cur_state.data(), (size_t)c->d * 8);
}

extern "C" int cssm_pmmh_run(cssm_pf* pf, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta,
                             const double* t, const double* y, const uint8_t* has_obs, size_t T, uint64_t seed,
                             size_t n_iters, double* ll, double* theta, int32_t* accepted, double* last_state) {
  if (!pf || !theta0 || !ll || !theta || !accepted || !last_state) return fail(CSSM_EINVAL_ARG, "null argument");
  const int d = cssm_pf_dim(pf);
  cssm_pmmh_chain* c = nullptr;
  int rc = cssm_pmmh_chain_create(desc, theta0, n_theta, delta, seed, d, &c);
  if (rc) return rc;
  std::vector<double> path((T + 1) * (size_t)d);
  for (size_t it = 0; it < n_iters && !rc; ++it) {
    uint64_t key = 0;
    const cssm_model_desc* pd = cssm_pmmh_chain_propose(c, it, &key);
    rc = cssm_pf_set_params(pf, pd);
    if (rc) break;
    cssm_pf_reseed(pf, key);
    double pll = 0.0;
    rc = cssm_pf_filter(pf, t, y, has_obs, T, &pll, nullptr, nullptr, path.data());   // state = pf(propParams)
    if (rc == CSSM_ENONFINITE) { pll = -cssm_inf(); rc = CSSM_OK; }          // a proposal the filter cannot weigh is rejected
    else if (rc) break;
    cssm_pmmh_chain_decide(c, it, pll, path.data() + T * (size_t)d, &ll[it], theta + it * n_theta, &accepted[it], last_state + it * (size_t)d);
  }
  cssm_pmmh_chain_destroy(c);
  return rc;
}

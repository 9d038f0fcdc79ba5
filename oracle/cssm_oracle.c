/*
 * cssm_oracle.c -- CPU restatement of the reference bootstrap particle filter.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; libcssm_pf never links or calls it.
 *
 * PARITY STATUS: "parity unpinned by the reference".  The reference
 * (jonnylaw/ComposableStateSpaceModels, Scala 2.13/JVM) can be neither compiled nor run in
 * the build image (no JVM, no jars), and its own tests pin no particle-filter number
 * (SamplingTest.scala:16-18 pins only the output LENGTH of
 * systematic resampling).  The restatement is therefore pinned by
 *   - scipy.stats golden vectors for every density (tests/golden/densities.json),
 *   - closed-form transition moments and the double-logistic OU quirk,
 *   - hand-worked systematic-resampling examples that follow Resampling.scala:52-72 literally,
 *   - the exact Kalman-filter log-likelihood for Brownian + Gaussian observation,
 *   - the reference's own property (output length == input length),
 *   - and (round 6) a SECOND, independent statement of the path: tests/golden/make_literal.py restates stepFilter, the
 *     transitions, f, the Poisson density, systematicResampling (sequential cumsum + searchsorted + the TreeMap's
 *     last-key-wins) and the LGCP step from the Scala lines in numpy / libm arithmetic without this file or
 *     include/cssm_numerics.h; its fixture (tests/golden/literal_runs.json) holds this file's literal mode to 1e-12 in
 *     ll with ESS and ancestors identical (tests/test_literal_mirror.py).  Only the variates are shared
 *     (oracle_pf_dump_normals, oracle_c_u): the reference's generators are unseeded.
 *
 * Each function cites the reference lines it follows as  model/<File>.scala:<lines>
 * (relative to src/main/scala/com/github/jonnylaw/).  Execution model is the reference's:
 * one thread, particle by particle, left fold over the data.
 *
 * Two arithmetic modes (flags of oracle_pf_create):
 *   contract (default)  random variates, exp/log/sincos and the weight sums follow
 *                       include/cssm_numerics.h, so a GPU run is comparable bit for bit.  The
 *                       fixed-point sums are formed here with the compiler's native
 *                       unsigned __int128 and long double, NOT with the header's two-limb
 *                       code, so the two implementations check each other.
 *   ORACLE_LITERAL_SUMS sums and the cumulative scan are the reference's sequential fp64
 *                       folds (Resampling.scala:21-24,57; ParticleFilter.scala:431-434,522-524);
 *   ORACLE_TIE_LAST     TreeMap duplicate-key semantics: on equal cumulative weights the LAST
 *                       particle wins (Resampling.scala:57, `tree ++ zip`), and a grid point
 *                       above the last cumulative weight is an error as in :41-42;
 *   ORACLE_LIBM         glibc exp/log/sin/cos instead of the contract polynomials (bounds the
 *                       effect of the contract's elementary functions on ll).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/cssm_numerics.h"
#include "../include/cssm_pf.h"

#define ORACLE_LITERAL_SUMS 1
#define ORACLE_TIE_LAST 2
#define ORACLE_LIBM 4
#define ORACLE_RESAMPLE_STRATIFIED 8    /* Resampling.stratifiedResampling, model/Resampling.scala:78-86 */
#define ORACLE_RESAMPLE_MULTINOMIAL 16  /* Resampling.multinomialResampling, model/Resampling.scala:92-96 */

#define ORACLE_OK 0
#define ORACLE_EINVAL (-1)
#define ORACLE_EEMPTY (-8)    /* findAllInTreeMap `.head` on an empty map (Resampling.scala:41-42) */
#define ORACLE_ENONFINITE (-5)

typedef unsigned __int128 u128;

/* One scalar latent component in Tree.flatten order (model/Tree.scala:49-53). */
typedef struct {
  int kind;        /* CSSM_SDE_* */
  int leaf;        /* leaf index */
  int idx;         /* index inside the leaf's DenseVector */
  int f_kind;      /* CSSM_F_* of the leaf */
  int period;
  double m0, c0;   /* c0 = exp(stored)            model/Sde.scala:72,101,136 */
  double mu;
  double phi;      /* OU: logistic(stored)        model/Sde.scala:136 */
  double sigma;    /* exp(stored)                 model/Sde.scala:72,101,136 */
} ocomp;

struct oracle_pf {
  int d, n_leaves, obs_kind, precision, flags, obs_df;
  double scale_sd;       /* exp(scale): Gaussian sd, NegBin size, Student-t v   model/Model.scala:147,171,211,244 */
  double ref_last, gmax_last; /* rescaling level and max log-weight of the last weighted step */
  double next_ref;       /* LGCP (contract v8, cssm_ref_predict): the level predicted for the NEXT weighted observation from the max of
                            the last one; NaN while none has run since the cloud was drawn or handed in (oracle_pf_set_particles) */
  double scale_raw;      /* ZIP: the stored scale v                              model/Model.scala:284,300 */
  ocomp comp[CSSM_MAX_DIM];
  uint64_t n, seed;
  uint64_t first, n_global; /* shard of a larger filter: global id of particle 0, global particle count */
  double t, ll;
  int32_t ess;
  uint32_t step;         /* observation index of the NEXT datum */
  int initialised;
  double* x;             /* resampled cloud, AoS [n][d]  (PfState.particles) */
  double* x1;            /* propagated cloud before resampling (x1 at ParticleFilter.scala:118) */
  double* w;             /* log-weights (ParticleFilter.scala:123) */
  double* w1;            /* exp(w - max) (:125) */
  double* C;             /* cumulative normalised weights (Resampling.scala:57) */
  uint32_t* anc;
};
typedef struct oracle_pf oracle_pf;

/* ------------------------------------------------------------------ elementary functions */

static double o_exp(const oracle_pf* pf, double x) { return (pf->flags & ORACLE_LIBM) ? exp(x) : cssm_exp(x); }
static double o_log(const oracle_pf* pf, double x) { return (pf->flags & ORACLE_LIBM) ? log(x) : cssm_log(x); }

/* pair number `pair` of stream (gid, step, tag): half pair & 1 of block pair >> 1 (include/cssm_numerics.h, counter layout) */
static void o_normal_pair(const oracle_pf* pf, uint64_t gid, uint32_t step, uint32_t tag, uint32_t pair, double* z0, double* z1) {
  if (pf->flags & ORACLE_LIBM) {
    const cssm_u32x4 blk = cssm_philox_draw(pf->seed, gid, step, tag, pair >> 1);
    const uint32_t a = (pair & 1u) ? blk.v[2] : blk.v[0], b = (pair & 1u) ? blk.v[3] : blk.v[1];
    double u1 = cssm_fma((double)a, 0x1.0p-32, (double)((b & 255u) + 1u) * 0x1.0p-40), u2 = (double)(b >> 8) * 0x1.0p-24;
    double r = sqrt(-2.0 * log(u1));
    *z0 = r * cos(6.283185307179586476925 * u2);
    *z1 = r * sin(6.283185307179586476925 * u2);
  } else {
    cssm_normal_pair_of(pf->seed, gid, step, tag, pair, CSSM_LOG_TAB, z0, z1);
  }
}

/* SdeParameter.logistic, model/SdeParameters.scala:214-216 */
static double o_logistic(const oracle_pf* pf, double x) { return 1.0 / (1.0 + o_exp(pf, -x)); }

/* ------------------------------------------------------------------ model construction */

/* Sde.buildParamRepeat, model/Sde.scala:177-179 */
static double rep(const double* v, int n, int i) { return v[i % n]; }

static int build_components(oracle_pf* pf, const cssm_model_desc* desc) {
  if (!desc || desc->n_leaves < 1 || desc->n_leaves > CSSM_MAX_LEAVES || !desc->leaves) return ORACLE_EINVAL;
  int d = 0;
  pf->n_leaves = desc->n_leaves;
  pf->obs_kind = desc->obs_kind;
  pf->precision = desc->lgcp_precision;
  for (int l = 0; l < desc->n_leaves; ++l) {
    const cssm_leaf_desc* L = &desc->leaves[l];
    if (L->dim < 1 || d + L->dim > CSSM_MAX_DIM) return ORACLE_EINVAL;
    if (L->n_m0 < 1 || L->n_c0 < 1 || L->n_sigma < 1 || !L->m0 || !L->c0 || !L->sigma) return ORACLE_EINVAL;
    if ((L->sde_kind == CSSM_SDE_GEN_BROWNIAN || L->sde_kind == CSSM_SDE_OU || L->sde_kind == CSSM_SDE_EULER_AFFINE) &&
        (L->n_mu < 1 || !L->mu)) return ORACLE_EINVAL;
    if ((L->sde_kind == CSSM_SDE_OU || L->sde_kind == CSSM_SDE_EULER_AFFINE) && (L->n_phi < 1 || !L->phi)) return ORACLE_EINVAL;
    if (L->f_kind == CSSM_F_SEASONAL && (L->dim != 2 * L->harmonics || L->period < 1)) return ORACLE_EINVAL;
    for (int i = 0; i < L->dim; ++i) {
      ocomp* c = &pf->comp[d + i];
      memset(c, 0, sizeof *c);
      c->kind = L->sde_kind; c->leaf = l; c->idx = i; c->f_kind = L->f_kind; c->period = L->period;
      c->m0 = rep(L->m0, L->n_m0, i);
      c->c0 = o_exp(pf, rep(L->c0, L->n_c0, i));              /* c.map(exp(_)) */
      switch (L->sde_kind) {
        case CSSM_SDE_BROWNIAN:                                /* model/Sde.scala:99-102 */
          c->sigma = o_exp(pf, rep(L->sigma, L->n_sigma, i));
          break;
        case CSSM_SDE_GEN_BROWNIAN:                            /* model/Sde.scala:70-73 */
          c->mu = rep(L->mu, L->n_mu, i);
          c->sigma = o_exp(pf, rep(L->sigma, L->n_sigma, i));
          break;
        case CSSM_SDE_OU:                                      /* model/Sde.scala:133-137 */
          c->phi = o_logistic(pf, rep(L->phi, L->n_phi, i));   /* applied to an already-logistic value: the double-logistic quirk */
          c->mu = rep(L->mu, L->n_mu, i);
          c->sigma = o_exp(pf, rep(L->sigma, L->n_sigma, i));
          break;
        case CSSM_SDE_EULER_AFFINE:                            /* user-defined Sde: raw a, b, g */
          c->mu = rep(L->mu, L->n_mu, i);
          c->phi = rep(L->phi, L->n_phi, i);
          c->sigma = rep(L->sigma, L->n_sigma, i);
          break;
        default: return ORACLE_EINVAL;
      }
    }
    d += L->dim;
  }
  pf->d = d;
  /* only the leftmost leaf supplies the observation model, model/Model.scala:118-120,132 */
  const cssm_leaf_desc* L0 = &desc->leaves[0];
  switch (desc->obs_kind) {
    case CSSM_OBS_GAUSSIAN: case CSSM_OBS_NEGBIN: case CSSM_OBS_STUDENT_T: case CSSM_OBS_ZIP:
      if (!L0->has_scale) return ORACLE_EINVAL;                /* "Must provide SD parameter" / "No scale parameter", model/Model.scala:150,179,250,294 */
      pf->scale_raw = L0->scale;
      pf->scale_sd = o_exp(pf, L0->scale);
      pf->obs_df = desc->obs_df;
      if (desc->obs_kind == CSSM_OBS_STUDENT_T && desc->obs_df < 1) return ORACLE_EINVAL;
      break;
    case CSSM_OBS_POISSON: case CSSM_OBS_LGCP: case CSSM_OBS_BERNOULLI: case CSSM_OBS_BETA: break;
    default: return ORACLE_EINVAL;
  }
  return ORACLE_OK;
}

int oracle_pf_create(const cssm_model_desc* desc, uint64_t n, uint64_t seed, int flags, oracle_pf** out) {
  if (!out || n < 1 || n >= 0xffffffffULL) return ORACLE_EINVAL;
  oracle_pf* pf = (oracle_pf*)calloc(1, sizeof *pf);
  if (!pf) return ORACLE_EINVAL;
  pf->flags = flags; pf->n = n; pf->seed = seed; pf->first = 0; pf->n_global = n;
  int rc = build_components(pf, desc);
  if (rc) { free(pf); return rc; }
  size_t nd = (size_t)n * pf->d;
  pf->x = (double*)malloc(nd * 8); pf->x1 = (double*)malloc(nd * 8);
  pf->w = (double*)malloc(n * 8); pf->w1 = (double*)malloc(n * 8); pf->C = (double*)malloc(n * 8);
  pf->anc = (uint32_t*)malloc(n * 4);
  for (uint64_t i = 0; i < n; ++i) pf->anc[i] = (uint32_t)i;
  *out = pf;
  return ORACLE_OK;
}

void oracle_pf_destroy(oracle_pf* pf) {
  if (!pf) return;
  free(pf->x); free(pf->x1); free(pf->w); free(pf->w1); free(pf->C); free(pf->anc); free(pf);
}

int oracle_pf_set_params(oracle_pf* pf, const cssm_model_desc* desc) {
  int d0 = pf->d;
  int rc = build_components(pf, desc);
  if (rc) return rc;
  return pf->d == d0 ? ORACLE_OK : ORACLE_EINVAL;
}
void oracle_pf_reseed(oracle_pf* pf, uint64_t seed) { pf->seed = seed; }

/* Effective (constrained) per-component parameters, for the K2 tests. out[d][5] = m0,c0,mu,phi,sigma */
int oracle_pf_components(const oracle_pf* pf, double* out) {
  for (int k = 0; k < pf->d; ++k) {
    out[5 * k + 0] = pf->comp[k].m0; out[5 * k + 1] = pf->comp[k].c0; out[5 * k + 2] = pf->comp[k].mu;
    out[5 * k + 3] = pf->comp[k].phi; out[5 * k + 4] = pf->comp[k].sigma;
  }
  return pf->d;
}

/* ------------------------------------------------------------------ random variates */

/* d standard normals of particle `gid` at (step, tag, substep): normal number q = substep*d + k is
 * element q&1 of Box-Muller pair q>>1 (include/cssm_numerics.h, counter layout). */
static void draw_normals(const oracle_pf* pf, uint64_t gid, uint32_t step, uint32_t tag, uint32_t sub, double* z) {
  for (int k = 0; k < pf->d; ++k) {
    uint32_t q = sub * (uint32_t)pf->d + (uint32_t)k;
    double z0, z1;
    o_normal_pair(pf, gid, step, tag, q >> 1, &z0, &z1);
    z[k] = (q & 1u) ? z1 : z0;
  }
}

/* The same for an ordinary step or the initial draw, where particles 2m and 2m+1 share stream m: particle gid owns
 * the normals q = (gid & 1) * d + k of that stream (include/cssm_numerics.h, counter layout). */
static void draw_normals_paired(const oracle_pf* pf, uint64_t gid, uint32_t step, uint32_t tag, double* z) {
  const uint32_t q0 = cssm_pair_first(gid, pf->d);
  for (int k = 0; k < pf->d; ++k) {
    uint32_t q = q0 + (uint32_t)k;
    double z0, z1;
    o_normal_pair(pf, cssm_pair_stream(gid), step, tag, q >> 1, &z0, &z1);
    z[k] = (q & 1u) ? z1 : z0;
  }
}

/* The injected random stream, dumped for the independent numpy statement of the path (tests/golden/make_literal.py: the reference draws
 * from an unseeded global generator -- what a second witness must share with this file are the VARIATES, not the arithmetic): the d
 * normals of every particle of this handle at (step, tag) in the handle's mode (ORACLE_LIBM: Box-Muller through libm) -- sub < 0: the
 * pair streams of an ordinary step / the initial draw; sub >= 0: sub-step `sub` of an LGCP event.  out[n][d]. */
void oracle_pf_dump_normals(const oracle_pf* pf, uint32_t step, int init, int sub, double* out) {
  const uint32_t tag = init ? CSSM_STREAM_INIT : CSSM_STREAM_STEP;
  for (uint64_t i = 0; i < pf->n; ++i) {
    if (sub < 0) draw_normals_paired(pf, pf->first + i, step, tag, out + i * pf->d);
    else draw_normals(pf, pf->first + i, step, tag, (uint32_t)sub, out + i * pf->d);
  }
}

/* ------------------------------------------------------------------ A1 initial state */

/* initialiseState, model/ParticleFilter.scala:105-108; initialState of the leaves:
 * Brownian model/Sde.scala:104-108, GenBrownian :75-80, OU :152-156, composed :206-209:
 * x0 = sqrt(c0) * z + m0 per component. */
int oracle_pf_init(oracle_pf* pf, double t0) {
  double z[CSSM_MAX_DIM];
  for (uint64_t i = 0; i < pf->n; ++i) {
    draw_normals_paired(pf, pf->first + i, 0, CSSM_STREAM_INIT, z);
    for (int k = 0; k < pf->d; ++k) pf->x[i * pf->d + k] = sqrt(pf->comp[k].c0) * z[k] + pf->comp[k].m0;
  }
  memcpy(pf->x1, pf->x, (size_t)pf->n * pf->d * 8);
  for (uint64_t i = 0; i < pf->n; ++i) pf->anc[i] = (uint32_t)i;
  pf->t = t0; pf->ll = 0.0; pf->ess = (int32_t)pf->n; pf->step = 0; pf->initialised = 1; pf->next_ref = cssm_nan();
  return ORACLE_OK;
}

/* FilterInit.initialiseState, model/ParticleFilter.scala:257-260 */
int oracle_pf_init_from(oracle_pf* pf, double t0, const double* state_d) {
  for (uint64_t i = 0; i < pf->n; ++i) memcpy(pf->x + i * pf->d, state_d, pf->d * 8);
  memcpy(pf->x1, pf->x, (size_t)pf->n * pf->d * 8);
  for (uint64_t i = 0; i < pf->n; ++i) pf->anc[i] = (uint32_t)i;
  pf->t = t0; pf->ll = 0.0; pf->ess = (int32_t)pf->n; pf->step = 0; pf->initialised = 1; pf->next_ref = cssm_nan();
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ A3 transitions */

/* One particle, all leaves left to right (composed stepFunction, model/Sde.scala:223-229). */
static void transition(const oracle_pf* pf, const double* x, double dt, const double* z, double* out) {
  for (int k = 0; k < pf->d; ++k) {
    const ocomp* c = &pf->comp[k];
    switch (c->kind) {
      case CSSM_SDE_BROWNIAN: {            /* model/Sde.scala:114-123: sqrt(sigma*dt) * z + x */
        double sd = sqrt(c->sigma * dt);
        out[k] = sd * z[k] + x[k];
        break;
      }
      case CSSM_SDE_GEN_BROWNIAN: {        /* model/Sde.scala:86-95: mean = x + mu*dt */
        double mean = x[k] + c->mu * dt;
        double sd = sqrt(c->sigma * dt);
        out[k] = sd * z[k] + mean;
        break;
      }
      case CSSM_SDE_OU: {                  /* model/Sde.scala:139-150 */
        double var = (c->sigma * c->sigma / (c->phi * 2.0)) * (1.0 - o_exp(pf, c->phi * -2.0 * dt));
        double mean = c->mu + (x[k] - c->mu) * o_exp(pf, -c->phi * dt);
        out[k] = sqrt(var) * z[k] + mean;
        break;
      }
      default: {                           /* stepEulerMaruyama, model/Sde.scala:30-43 */
        double dW = sqrt(dt) * z[k];
        double a = (c->mu + c->phi * x[k]) * dt;
        double b = c->sigma * dW;
        out[k] = (x[k] + a) + b;
        break;
      }
    }
  }
}

/* ------------------------------------------------------------------ A4 linear map f */

/* f(s, t): composed sum model/Model.scala:122-128 (left-nested), leaf "first component"
 * :271, seasonal buildF/f :217-225 with the contract's period reduction. */
static double gamma_of(const oracle_pf* pf, const double* x, double t) {
  double g = 0.0, acc = 0.0;
  int cur = -1, nleaf = 0;
  for (int k = 0; k < pf->d; ++k) {
    const ocomp* c = &pf->comp[k];
    if (c->leaf != cur) {
      if (cur >= 0) { g = (nleaf == 0) ? acc : g + acc; ++nleaf; }
      cur = c->leaf; acc = 0.0;
    }
    if (c->f_kind == CSSM_F_FIRST) {
      if (c->idx == 0) acc = x[k];
    } else {
      double a = (double)(c->idx / 2 + 1), sn, cs;
      if (pf->flags & ORACLE_LIBM) {
        double om = 2.0 * M_PI / c->period;                    /* literal: cos(frequency * a * t) */
        cs = cos(om * a * t); sn = sin(om * a * t);
      } else {
        cssm_sincos2pi(cssm_seasonal_phase(a, t, (double)c->period), &sn, &cs);
      }
      double term = ((c->idx & 1) ? sn : cs) * x[k];
      acc = (c->idx == 0) ? term : acc + term;                 /* F(t) dot x, left to right */
    }
  }
  g = (nleaf == 0) ? acc : g + acc;
  return g;
}

/* ------------------------------------------------------------------ A5 observation densities */

/* breeze Poisson(lambda).logProbabilityOf(k) = -lambda + k*log(lambda) - lgamma(k+1) with
 * lambda = exp(gamma) (model/Model.scala:269,273).  k*log(exp(gamma)) is evaluated as
 * k*gamma (<= 1 ulp*k apart; no JVM run exists to bit-match). */
double oracle_logdens_poisson(double gamma, double y) {
  long long k = (long long)y;                                  /* y.toInt truncation */
  return -cssm_exp(gamma) + (double)k * gamma - cssm_lgamma_kp1(k);
}
/* breeze Gaussian(mu, sd).logPdf(y) = -((y-mu)/sd)^2/2 - log(sqrt(2 pi) * sd), model/Model.scala:252-258 */
double oracle_logdens_gaussian(double gamma, double y, double sd) {
  double dd = (y - gamma) / sd;
  return -(dd * dd) / 2.0 - cssm_log(2.5066282746310002 * sd);
}

/* NegativeBinomialModel.dataLikelihood, model/Model.scala:186-195 (size = exp(scale), mu = exp(gamma)) */
double oracle_logdens_negbin(double gamma, double y, double size) {
  long long k = (long long)y;
  double mu = cssm_exp(gamma);
  return (cssm_lgamma(size + (double)k) - cssm_lgamma_kp1(k) - cssm_lgamma(size)) + size * cssm_log(size / (mu + size)) +
         (double)k * cssm_log(mu / (mu + size));
}
/* ZeroInflatedPoisson.dataLikelihood, model/Model.scala:298-307 (v = stored scale) */
double oracle_logdens_zip(double gamma, double y, double v) {
  long long k = (long long)y;
  double ev = cssm_exp(v);
  double p = ev / (1.0 + ev);
  if (k == 0) return cssm_log(p + (1.0 - p) * cssm_exp(-cssm_exp(gamma)));
  return ((-cssm_log(1.0 + ev) + (double)k * gamma) - cssm_exp(gamma)) - cssm_lgamma_kp1(k);
}
/* BernoulliModel.link / dataLikelihood, model/Model.scala:318-336 */
double oracle_logdens_bernoulli(double gamma, double y) {
  double link = (gamma > 6.0) ? 1.0 : ((gamma < -6.0) ? 0.0 : 1.0 / (1.0 + cssm_exp(-gamma)));
  if (y == 1.0) return (link == 0.0) ? -1e99 : cssm_log(link);
  return (link == 1.0) ? -1e99 : cssm_log(1.0 - link);
}
/* StudentsTModel.dataLikelihood, model/Model.scala:155-160: 1/v * StudentsT(df).logPdf((y - eta)/v)
 * (the 1/v factor multiplies the LOG-pdf, as written); breeze StudentsT.logPdf(x) =
 * lgamma((df+1)/2) - lgamma(df/2) - log(pi df)/2 - (df+1)/2 log(1 + x^2/df) */
double oracle_logdens_student_t(double gamma, double y, double v, int df) {
  double d = (double)df;
  double x = (y - gamma) / v;
  double c0 = cssm_lgamma((d + 1.0) / 2.0) - cssm_lgamma(d / 2.0) - 0.5 * cssm_log(3.14159265358979311600 * d);
  return (1.0 / v) * (c0 - ((d + 1.0) / 2.0) * cssm_log(1.0 + (x * x) / d));
}
/* BetaModel.dataLikelihood, model/Model.scala:349-352: Beta(exp(-gamma), 1).logPdf(y) = (a-1) log y - logB(a,1),
 * and logB(a,1) = lgamma(a) + lgamma(1) - lgamma(a+1) = -log a = gamma: evaluated in that closed form. */
double oracle_logdens_beta(double gamma, double y) { return (cssm_exp(-gamma) - 1.0) * cssm_log(y) - gamma; }

static double logdens(const oracle_pf* pf, double gamma, double y) {
  switch (pf->obs_kind) {
    case CSSM_OBS_NEGBIN: return oracle_logdens_negbin(gamma, y, pf->scale_sd);
    case CSSM_OBS_ZIP: return oracle_logdens_zip(gamma, y, pf->scale_raw);
    case CSSM_OBS_BERNOULLI: return oracle_logdens_bernoulli(gamma, y);
    case CSSM_OBS_STUDENT_T: return oracle_logdens_student_t(gamma, y, pf->scale_sd, pf->obs_df);
    case CSSM_OBS_BETA: return oracle_logdens_beta(gamma, y);
    default: break;
  }
  if (pf->obs_kind == CSSM_OBS_POISSON) {
    long long k = (long long)y;
    return -o_exp(pf, gamma) + (double)k * gamma - cssm_lgamma_kp1(k);
  }
  double dd = (y - gamma) / pf->scale_sd;
  return -(dd * dd) / 2.0 - o_log(pf, 2.5066282746310002 * pf->scale_sd);
}

/* ------------------------------------------------------------------ A8 systematic resampling */

static u128 fix_from_double(double w) {                         /* floor(w * 2^96), independent of the header */
  if (!(w > 0.0) || isinf(w)) return 0;
  long double s = floorl((long double)w * 0x1p96L);             /* 64-bit mantissa: the product is exact */
  return (u128)s;
}

/*
 * Resampling.systematicResampling, model/Resampling.scala:63-72, with treeEcdf :52-58,
 * normalise :21-24 and findAllInTreeMap :36-46.  `findAllInTreeMap` walks the grid points in
 * increasing order and takes the first key >= k, which is the two-pointer merge below.
 */
int oracle_resample_systematic(const double* w, uint64_t n, double u, uint32_t* anc, double* C_out, int flags) {
  double* C = C_out ? C_out : (double*)malloc(n * 8);
  if (flags & ORACLE_LITERAL_SUMS) {
    double total = 0.0;
    for (uint64_t i = 0; i < n; ++i) total = total + w[i];     /* prob.foldLeft(0.0)(_ + _) */
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) { acc = acc + w[i] / total; C[i] = acc; }  /* scanLeft(0.0)(_ + _).drop(1) */
  } else {
    u128 tot = 0, acc = 0;
    for (uint64_t i = 0; i < n; ++i) tot += fix_from_double(w[i]);
    double totd = (double)tot;                                  /* compiler's RNE u128 -> double */
    for (uint64_t i = 0; i < n; ++i) { acc += fix_from_double(w[i]); C[i] = (double)acc / totd; }
  }
  int rc = ORACLE_OK;
  uint64_t j = 0;
  double nd = (double)n;
  for (uint64_t i = 0; i < n; ++i) {
    double k = (u + (double)i) / nd;                            /* ks, Resampling.scala:69 */
    while (j < n && C[j] < k) ++j;                              /* remMap.from(k) */
    if (j >= n) {
      if (flags & ORACLE_TIE_LAST) { rc = ORACLE_EEMPTY; break; }   /* `.head` of an empty map */
      j = n - 1;                                                /* contract: clamp (cannot occur: C[n-1] == 1) */
    }
    uint64_t a = j;
    if (flags & ORACLE_TIE_LAST) while (a + 1 < n && C[a + 1] == C[a]) ++a;  /* duplicate key: last insert wins */
    anc[i] = (uint32_t)a;
  }
  if (!C_out) free(C);
  return rc;
}

/* Contract cumulative weights C_j = RN(RN(S_j)/RN(S_tot)) of the fixed-point sums. */
static void contract_cumw(const double* w, uint64_t n, double* C) {
  u128 tot = 0, acc = 0;
  for (uint64_t i = 0; i < n; ++i) tot += fix_from_double(w[i]);
  double totd = (double)tot;
  for (uint64_t i = 0; i < n; ++i) { acc += fix_from_double(w[i]); C[i] = (double)acc / totd; }
}

/* Resampling.stratifiedResampling, model/Resampling.scala:78-86: ks_i = (i + nextDouble)/n, one uniform per slot
 * (ordered, so findAllInTreeMap is again the two-pointer merge). */
int oracle_resample_stratified(const double* w, uint64_t n, uint64_t seed, uint32_t step, uint32_t* anc, double* C_out) {
  double* C = C_out ? C_out : (double*)malloc(n * 8);
  contract_cumw(w, n, C);
  uint64_t j = 0;
  double nd = (double)n;
  for (uint64_t i = 0; i < n; ++i) {
    double k = cssm_strat_grid(seed, step, i, nd);
    while (j < n - 1 && C[j] < k) ++j;
    anc[i] = (uint32_t)j;
  }
  if (!C_out) free(C);
  return ORACLE_OK;
}

/* Resampling.multinomialResampling, model/Resampling.scala:92-96: n independent draws of breeze Multinomial(weights),
 * each the first index whose cumulative weight reaches u * sum; output in draw order. */
int oracle_resample_multinomial(const double* w, uint64_t n, uint64_t seed, uint32_t step, uint32_t* anc, double* C_out) {
  double* C = C_out ? C_out : (double*)malloc(n * 8);
  contract_cumw(w, n, C);
  for (uint64_t i = 0; i < n; ++i) {
    double ui = cssm_multi_uniform(seed, step, i);
    uint64_t j = 0;
    while (j < n - 1 && C[j] < ui) ++j;                      /* linear, as Multinomial.draw scans */
    anc[i] = (uint32_t)j;
  }
  if (!C_out) free(C);
  return ORACLE_OK;
}

/* EXTENSION (not a restatement of running reference code): residual resampling in the intent its scaladoc states
 * (model/Resampling.scala:124-129: "particle (xi, wi) appears ki = n * wi times; resample m = n - total allocated particles
 * according to w = n * wi - ki using other resampling technique") -- the body at :130-146 cannot run (it indexes with draws of
 * Vector.range(1, m) and exp-normalises weights that are already exponentiated).  The twin of cssm_resample_residual
 * (csrc/cssm_residual.hip), whose header states the arithmetic: p_i = RN(RN(F_i) / RN(S)) * n, k_i = floor(p_i) copies in particle
 * order, then m multinomial draws on the residuals r_i = p_i - k_i (contract sums, cssm_multi_uniform(seed, step, t)). */
int oracle_resample_residual(const double* w, uint64_t n, uint64_t seed, uint32_t step, uint32_t* anc) {
  u128 tot = 0;
  for (uint64_t i = 0; i < n; ++i) tot += fix_from_double(w[i]);
  if (tot == 0) return ORACLE_EEMPTY;
  const double totd = (double)tot, nd = (double)n;
  double* r = (double*)malloc(n * 8);
  uint64_t K = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const double p = ((double)fix_from_double(w[i]) / totd) * nd;
    const uint64_t k = (p >= nd) ? n : (uint64_t)p;
    r[i] = p - (double)k;
    if (K + k > n) { free(r); return ORACLE_EEMPTY; }
    for (uint64_t c = 0; c < k; ++c) anc[K++] = (uint32_t)i;
  }
  if (K < n) {
    double* C = (double*)malloc(n * 8);
    u128 rt = 0;
    for (uint64_t i = 0; i < n; ++i) rt += fix_from_double(r[i]);
    if (rt == 0) { free(C); free(r); return ORACLE_EEMPTY; }
    contract_cumw(r, n, C);
    const uint64_t m = n - K;
    for (uint64_t t = 0; t < m; ++t) {
      const double ut = cssm_multi_uniform(seed, step, t);
      uint64_t lo = 0, hi = n - 1;                               /* first j with C_j >= u_t (C is non-decreasing and ends at 1) */
      while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (C[mid] >= ut) hi = mid; else lo = mid + 1; }
      anc[K + t] = (uint32_t)lo;
    }
    free(C);
  }
  free(r);
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ A6/A7 sums, ll, ess */

/* Tail of stepFilter after the log-weights exist: model/ParticleFilter.scala:124-130
 * (and :218-225 for LGCP, same arithmetic). */
static int weigh_and_resample(oracle_pf* pf, double y) {
  uint64_t n = pf->n;
  int d = pf->d;
  double max = -INFINITY;
  for (uint64_t i = 0; i < n; ++i) {
    if (pf->w[i] != pf->w[i]) return ORACLE_ENONFINITE;
    if (pf->w[i] > max) max = pf->w[i];                         /* w.max */
  }
  if (max == -INFINITY || isinf(max)) return ORACLE_ENONFINITE;
  /* The contract rescales by the reference level of the observation when the max allows it (cssm_numerics.h,
   * cssm_ref_level / cssm_ref_choose); ORACLE_LITERAL_SUMS keeps the reference's own w - max. */
  if (!(pf->flags & ORACLE_LITERAL_SUMS)) {
    /* LGCP: gamma - hazard has no bound known in advance; its level is predicted from the max of the weighted observation
     * before (cssm_ref_predict), NaN -- i.e. the max -- where there is none */
    double c = (pf->obs_kind == CSSM_OBS_LGCP) ? pf->next_ref : cssm_ref_level(pf->obs_kind, y, pf->scale_sd, (double)pf->obs_df);
    pf->gmax_last = max;
    pf->next_ref = cssm_ref_predict(max);
    max = cssm_ref_choose(c, max);
  }
  pf->ref_last = max;
  for (uint64_t i = 0; i < n; ++i) pf->w1[i] = o_exp(pf, pf->w[i] - max);   /* :125 */
  const cssm_u32x4 bu = cssm_philox_draw(pf->seed, 0, pf->step, CSSM_STREAM_U, 0);
  double u = cssm_u01(bu.v[0], bu.v[1]);
  int rc;
  if (pf->flags & ORACLE_RESAMPLE_STRATIFIED) rc = oracle_resample_stratified(pf->w1, n, pf->seed, pf->step, pf->anc, pf->C);
  else if (pf->flags & ORACLE_RESAMPLE_MULTINOMIAL) rc = oracle_resample_multinomial(pf->w1, n, pf->seed, pf->step, pf->anc, pf->C);
  else rc = oracle_resample_systematic(pf->w1, n, u, pf->anc, pf->C, pf->flags);   /* :126 */
  if (rc) return rc;
  if (pf->flags & ORACLE_LITERAL_SUMS) {
    double sum = 0.0;
    for (uint64_t i = 0; i < n; ++i) sum = sum + pf->w1[i];     /* ParticleFilter.mean, :522-524 */
    pf->ll = pf->ll + max + o_log(pf, sum / (double)n);         /* :127 */
    double total = 0.0, s2 = 0.0;
    for (uint64_t i = 0; i < n; ++i) total = total + pf->w1[i]; /* Resampling.normalise */
    for (uint64_t i = 0; i < n; ++i) { double q = pf->w1[i] / total; s2 = s2 + q * q; }
    pf->ess = (int32_t)floor(1.0 / s2);                         /* :431-434 */
  } else {
    u128 S = 0, S2 = 0;
    for (uint64_t i = 0; i < n; ++i) { S += fix_from_double(pf->w1[i]); S2 += fix_from_double(pf->w1[i] * pf->w1[i]); }
    double tot = (double)S * 0x1p-96, tot2 = (double)S2 * 0x1p-96;
    pf->ll = pf->ll + max + o_log(pf, tot / (double)n);
    pf->ess = (int32_t)floor(1.0 / (tot2 / (tot * tot)));
  }
  for (uint64_t i = 0; i < n; ++i) memcpy(pf->x + i * d, pf->x1 + (uint64_t)pf->anc[i] * d, d * 8);   /* :130 */
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ A2 stepFilter */

/* stepFilter, model/ParticleFilter.scala:116-132 */
static int step_generic(oracle_pf* pf, double t, double y, int has_obs) {
  double z[CSSM_MAX_DIM];
  int d = pf->d;
  double dt = t - pf->t;                                        /* :117 */
  for (uint64_t i = 0; i < pf->n; ++i) {                        /* :118 */
    draw_normals_paired(pf, pf->first + i, pf->step, CSSM_STREAM_STEP, z);
    transition(pf, pf->x + i * d, dt, z, pf->x1 + i * d);
  }
  if (!has_obs) {                                               /* :121 */
    memcpy(pf->x, pf->x1, (size_t)pf->n * d * 8);
    for (uint64_t i = 0; i < pf->n; ++i) pf->anc[i] = (uint32_t)i;
    pf->t = t; pf->step++;
    return ORACLE_OK;
  }
  for (uint64_t i = 0; i < pf->n; ++i) pf->w[i] = logdens(pf, gamma_of(pf, pf->x1 + i * d, t), y);   /* :123 */
  int rc = weigh_and_resample(pf, y);
  if (rc) return rc;
  pf->t = t; pf->step++;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ A10 LGCP step */

/* FilterLgcp.stepFilter / calcWeight, model/ParticleFilter.scala:184-226, with
 * Sde.simInit / simInitStream, model/Sde.scala:57-66. */
static int step_lgcp(oracle_pf* pf, double t) {
  double z[CSSM_MAX_DIM], xs[CSSM_MAX_DIM], xn[CSSM_MAX_DIM];
  int d = pf->d;
  double dt = t - pf->t;                                        /* :211 */
  if (dt == 0) {                                                /* :212-213: (x, f, f) -> weight f - f */
    for (uint64_t i = 0; i < pf->n; ++i) {
      memcpy(pf->x1 + i * d, pf->x + i * d, d * 8);
      double g = gamma_of(pf, pf->x + i * d, t);
      pf->w[i] = g - g;
    }
  } else {
    double delta = pow(10.0, -pf->precision);
    uint32_t nsub = (uint32_t)ceil(dt / delta);                 /* :190 */
    for (uint64_t i = 0; i < pf->n; ++i) {
      memcpy(xs, pf->x + i * d, d * 8);
      double tau = t, haz = 0.0;                                /* simInit starts the clock at y.t, :194,215 */
      for (uint32_t k = 0; k < nsub; ++k) {
        draw_normals(pf, pf->first + i, pf->step, CSSM_STREAM_STEP, k, z);
        transition(pf, xs, delta, z, xn);                       /* x <- stepFunction(dt)(s.state), Sde.scala:59 */
        memcpy(xs, xn, d * 8);
        tau = tau + delta;                                      /* t = s.time + dt, Sde.scala:60 */
        haz = haz + o_exp(pf, gamma_of(pf, xs, tau)) * delta;   /* :203-205 */
      }
      memcpy(pf->x1 + i * d, xs, d * 8);
      pf->w[i] = gamma_of(pf, xs, t) - haz;                     /* :200, :217 */
    }
  }
  int rc = weigh_and_resample(pf, 0.0);                         /* :218-223 */
  if (rc) return rc;
  pf->t = t; pf->step++;
  return ORACLE_OK;
}

int oracle_pf_step(oracle_pf* pf, double t, double y, int has_obs, double* ll_out, int32_t* ess_out) {
  if (!pf->initialised) return ORACLE_EINVAL;
  int rc = (pf->obs_kind == CSSM_OBS_LGCP) ? step_lgcp(pf, t) : step_generic(pf, t, y, has_obs);
  if (ll_out) *ll_out = pf->ll;
  if (ess_out) *ess_out = pf->ess;
  return rc;
}

/* ------------------------------------------------------------------ shard helpers (tests of the multi-rank orchestration) */

/* This handle holds global particles [first, first + n) of an n_global-particle filter: variates
 * are keyed by the global id.  Only propagate_only / set_particles are meaningful on a shard. */
void oracle_pf_set_shard(oracle_pf* pf, uint64_t first, uint64_t n_global) { pf->first = first; pf->n_global = n_global; }

/* Lines :118 and :123 of stepFilter only (LGCP: calcWeight): x1 and the log-weights, no resampling. */
int oracle_pf_propagate_only(oracle_pf* pf, double t, double y, int has_obs) {
  double z[CSSM_MAX_DIM], xs[CSSM_MAX_DIM], xn[CSSM_MAX_DIM];
  int d = pf->d;
  double dt = t - pf->t;
  if (pf->obs_kind == CSSM_OBS_LGCP) {
    double delta = pow(10.0, -pf->precision);
    uint32_t nsub = dt == 0 ? 0 : (uint32_t)ceil(dt / delta);
    for (uint64_t i = 0; i < pf->n; ++i) {
      memcpy(xs, pf->x + i * d, d * 8);
      double tau = t, haz = 0.0;
      for (uint32_t k = 0; k < nsub; ++k) {
        draw_normals(pf, pf->first + i, pf->step, CSSM_STREAM_STEP, k, z);
        transition(pf, xs, delta, z, xn);
        memcpy(xs, xn, d * 8);
        tau = tau + delta;
        haz = haz + o_exp(pf, gamma_of(pf, xs, tau)) * delta;
      }
      memcpy(pf->x1 + i * d, xs, d * 8);
      double g = gamma_of(pf, xs, t);
      pf->w[i] = nsub ? g - haz : g - g;
    }
  } else {
    for (uint64_t i = 0; i < pf->n; ++i) {
      draw_normals_paired(pf, pf->first + i, pf->step, CSSM_STREAM_STEP, z);
      transition(pf, pf->x + i * d, dt, z, pf->x1 + i * d);
      if (has_obs) pf->w[i] = logdens(pf, gamma_of(pf, pf->x1 + i * d, t), y);
    }
  }
  pf->t = t; pf->step++;
  return ORACLE_OK;
}

/* Overwrite the current cloud (SoA in, [d][n]); n may not change. */
void oracle_pf_set_particles(oracle_pf* pf, const double* soa) {
  pf->next_ref = cssm_nan();   /* (a host resampler's cloud: no native observation whose max could predict the next level) */
  for (uint64_t i = 0; i < pf->n; ++i) for (int k = 0; k < pf->d; ++k) pf->x[i * pf->d + k] = soa[(uint64_t)k * pf->n + i];
}

/* ------------------------------------------------------------------ A9 drivers */

/* Resampling.sampleOne, model/Resampling.scala:151-154: abs(nextInt) % size */
static uint64_t pick_index(const oracle_pf* pf, uint32_t s) {
  int32_t r = (int32_t)cssm_philox_draw(pf->seed, 0, s, CSSM_STREAM_PICK, 0).v[0];
  uint32_t a = r < 0 ? (uint32_t)0 - (uint32_t)r : (uint32_t)r;
  return (uint64_t)a % pf->n;
}

/* the same index for a cloud of n particles under `seed` (tests/oracle_shard.py: which rank owns the pick) */
uint64_t oracle_c_pick(uint64_t seed, uint32_t s, uint64_t n) {
  int32_t r = (int32_t)cssm_philox_draw(seed, 0, s, CSSM_STREAM_PICK, 0).v[0];
  uint32_t a = r < 0 ? (uint32_t)0 - (uint32_t)r : (uint32_t)r;
  return (uint64_t)a % n;
}

/* llFilter model/ParticleFilter.scala:137-140 and filter :152-158 (path != NULL) */
int oracle_pf_filter(oracle_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T,
                     double* ll_out, double* ll_t, int32_t* ess_t, double* path) {
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];      /* data.minBy(_.t).t */
  int rc = oracle_pf_init(pf, t0);
  if (rc) return rc;
  if (path) memcpy(path, pf->x + pick_index(pf, 0) * pf->d, pf->d * 8);
  for (size_t s = 0; s < T; ++s) {
    rc = oracle_pf_step(pf, t[s], y[s], has ? has[s] : 1, NULL, NULL);
    if (rc) return rc;
    if (ll_t) ll_t[s] = pf->ll;
    if (ess_t) ess_t[s] = pf->ess;
    if (path) memcpy(path + (s + 1) * pf->d, pf->x + pick_index(pf, (uint32_t)(s + 1)) * pf->d, pf->d * 8);
  }
  if (ll_out) *ll_out = pf->ll;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------ getters (SoA out) */

uint64_t oracle_pf_num_particles(const oracle_pf* pf) { return pf->n; }
int oracle_pf_dim(const oracle_pf* pf) { return pf->d; }
void oracle_pf_get_particles(const oracle_pf* pf, double* out) {
  for (uint64_t i = 0; i < pf->n; ++i) for (int k = 0; k < pf->d; ++k) out[(uint64_t)k * pf->n + i] = pf->x[i * pf->d + k];
}
void oracle_pf_get_proposed(const oracle_pf* pf, double* out) {
  for (uint64_t i = 0; i < pf->n; ++i) for (int k = 0; k < pf->d; ++k) out[(uint64_t)k * pf->n + i] = pf->x1[i * pf->d + k];
}
void oracle_pf_get_logw(const oracle_pf* pf, double* out) { memcpy(out, pf->w, pf->n * 8); }
void oracle_pf_get_ancestors(const oracle_pf* pf, uint32_t* out) { memcpy(out, pf->anc, pf->n * 4); }
void oracle_pf_get_cumw(const oracle_pf* pf, double* out) { memcpy(out, pf->C, pf->n * 8); }
/* contract pass-throughs for the shard test double */
double oracle_pf_ref_level(const oracle_pf* pf, double y) {
  return (pf->obs_kind == CSSM_OBS_LGCP) ? pf->next_ref : cssm_ref_level(pf->obs_kind, y, pf->scale_sd, (double)pf->obs_df);
}
double oracle_c_ref_choose(double c, double max) { return cssm_ref_choose(c, max); }
double oracle_c_ref_predict(double prev_max) { return cssm_ref_predict(prev_max); }
uint64_t oracle_c_strat_count(double C, uint64_t seed, uint32_t step, uint64_t n) { return cssm_strat_count(C, seed, step, n); }
uint64_t oracle_c_order_key(double x) { return cssm_order_key(x); }
double oracle_c_order_unkey(uint64_t k) { return cssm_order_unkey(k); }
void oracle_pf_get_ref(const oracle_pf* pf, double* ref, double* gmax) { *ref = pf->ref_last; *gmax = pf->gmax_last; }

/* ------------------------------------------------------------------ cloud summaries */

/* link of the observing leaf: model/Model.scala:24 (identity), :183,269,296 (exp), :318-326 (Bernoulli), :345 (Beta) */
static double o_link(const oracle_pf* pf, double g) {
  switch (pf->obs_kind) {
    case CSSM_OBS_POISSON: case CSSM_OBS_NEGBIN: case CSSM_OBS_ZIP: return cssm_exp(g);
    case CSSM_OBS_BERNOULLI: return (g > 6.0) ? 1.0 : ((g < -6.0) ? 0.0 : 1.0 / (1.0 + cssm_exp(-g)));
    case CSSM_OBS_BETA: return cssm_exp(-g);
    default: return g;
  }
}
static long long clamp_rank(long long r, uint64_t n) { return r < 0 ? 0 : (r > (long long)n - 1 ? (long long)n - 1 : r); }
static int cmp_double(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }

/* getIntervals, model/ParticleFilter.scala:415-424: meanState (:465-479, weights 1/N), getallCredibleIntervals
 * (:488-512: sorted(N - index - 1), sorted(index - 1)), getOrderStatistic of eta (:455-460: sorted(N - index),
 * sorted(index)), meanEta = link(f(stateMean, t)). */
static int summary_of(const oracle_pf* pf, const double* x, double time, double interval, double* mean, double* lower, double* upper,
                      double* eta_of_mean, double* eta_lower, double* eta_upper) {
  uint64_t n = pf->n;
  int d = pf->d;
  double* col = (double*)malloc(n * 8);
  double w = 1.0 / (double)n;                                   /* normalisedWeights = w / w.sum */
  long long idx = (long long)floor(interval * (double)n);
  for (int k = 0; k < d; ++k) {
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) { col[i] = x[i * d + k]; acc = acc + col[i] * w; }
    mean[k] = acc;
    qsort(col, n, 8, cmp_double);
    lower[k] = col[clamp_rank((long long)n - idx - 1, n)];
    upper[k] = col[clamp_rank(idx - 1, n)];
  }
  for (uint64_t i = 0; i < n; ++i) col[i] = o_link(pf, gamma_of(pf, x + i * d, time));
  qsort(col, n, 8, cmp_double);
  *eta_lower = col[clamp_rank((long long)n - idx, n)];
  *eta_upper = col[clamp_rank(idx, n)];
  *eta_of_mean = o_link(pf, gamma_of(pf, mean, time));
  free(col);
  return ORACLE_OK;
}

int oracle_pf_summary(const oracle_pf* pf, double interval, double* mean, double* lower, double* upper,
                      double* eta_of_mean, double* eta_lower, double* eta_upper) {
  return summary_of(pf, pf->x, pf->t, interval, mean, lower, upper, eta_of_mean, eta_lower, eta_upper);
}

/* eta = link(f(x, t)) of every particle of the current cloud (out[n]) and of one given state: the pieces of summary_of a
 * sharded summary needs separately (tests/oracle_shard.py: the order statistics are global ranks, found over the shards). */
void oracle_pf_eta(const oracle_pf* pf, double* out) {
  for (uint64_t i = 0; i < pf->n; ++i) out[i] = o_link(pf, gamma_of(pf, pf->x + i * pf->d, pf->t));
}
double oracle_pf_eta_of(const oracle_pf* pf, const double* state) { return o_link(pf, gamma_of(pf, state, pf->t)); }

/* ------------------------------------------------------------------ FilterInterpolate */

/* FilterInterpolate.stepInterpolate / filterInterpolate, model/ParticleFilter.scala:273-311, followed by the
 * summary of examples/Interpolate.scala:42-44.  Literal: every particle IS its path (newest state first in the
 * reference's List; here oldest first in an array), a weighted step copies whole paths by the resampled indices.
 * Output entry k (k = 0..T): getIntervals of { path_i[src] : i } where src = k, evaluated with the time of index k
 * -- or, with reference_pairing != 0, src = T - k: what `(interpolated, interpolated.last.particles.transpose)
 * .zipped` really pairs, because the transposed paths run newest-first (an evident slip of the example that
 * only a drop-in needs to reproduce). */
int oracle_pf_interpolate(oracle_pf* pf, const double* t, const double* y, const uint8_t* has, size_t T, double interval,
                          int reference_pairing, double* ll_out, double* mean, double* lower, double* upper,
                          double* eta_of_mean, double* eta_lower, double* eta_upper) {
  if (pf->obs_kind == CSSM_OBS_LGCP) return ORACLE_EINVAL;
  uint64_t n = pf->n;
  int d = pf->d;
  size_t L = (T + 1) * (size_t)d;
  double t0 = t[0];
  for (size_t s = 1; s < T; ++s) if (t[s] < t0) t0 = t[s];
  int rc = oracle_pf_init(pf, t0);                               /* x0 = initialState.sample(particles) map (_ :: Nil) */
  if (rc) return rc;
  double* paths = (double*)calloc(n * L, 8), *next = (double*)calloc(n * L, 8), *cloud = (double*)malloc(n * d * 8);
  for (uint64_t i = 0; i < n; ++i) memcpy(paths + i * L, pf->x + i * d, d * 8);
  for (size_t s = 0; s < T && !rc; ++s) {
    rc = oracle_pf_step(pf, t[s], y[s], has ? has[s] : 1, NULL, NULL);
    if (rc) break;
    /* x1 = step(x.head) :: x, then resample(x1, w1): path i becomes (path anc[i]) ++ x1[anc[i]]; pf->x[i] = x1[anc[i]] */
    for (uint64_t i = 0; i < n; ++i) {
      memcpy(next + i * L, paths + (uint64_t)pf->anc[i] * L, (s + 1) * d * 8);
      memcpy(next + i * L + (s + 1) * d, pf->x + i * d, d * 8);
    }
    double* sw = paths; paths = next; next = sw;
  }
  if (!rc) {
    if (ll_out) *ll_out = pf->ll;
    for (size_t k = 0; k <= T; ++k) {
      size_t src = reference_pairing ? T - k : k;
      double time = k == 0 ? t0 : t[k - 1];
      for (uint64_t i = 0; i < n; ++i) memcpy(cloud + i * d, paths + i * L + src * d, d * 8);
      summary_of(pf, cloud, time, interval, mean + k * d, lower + k * d, upper + k * d, eta_of_mean + k, eta_lower + k, eta_upper + k);
    }
  }
  free(paths); free(next); free(cloud);
  return rc;
}

/* ------------------------------------------------------------------ A11 PMMH */

/* Parameters.flattenParams order, model/Parameters.scala:88-95; SdeParameter.flatten,
 * model/SdeParameters.scala:73,111,153. Returns the count; writes pointers to every stored
 * scalar of a MUTABLE copy of the descriptor. */
static size_t theta_slots(cssm_model_desc* desc, double** slots, size_t cap) {
  size_t n = 0;
  for (int l = 0; l < desc->n_leaves; ++l) {
    cssm_leaf_desc* L = (cssm_leaf_desc*)&desc->leaves[l];
#define PUSH(ptr, cnt) for (int i = 0; i < (cnt); ++i) { if (n < cap) slots[n] = (double*)&(ptr)[i]; ++n; }
    if (L->has_scale) { if (n < cap) slots[n] = &L->scale; ++n; }
    PUSH(L->m0, L->n_m0) PUSH(L->c0, L->n_c0)
    if (L->sde_kind == CSSM_SDE_GEN_BROWNIAN) { PUSH(L->mu, L->n_mu) }
    else if (L->sde_kind == CSSM_SDE_OU || L->sde_kind == CSSM_SDE_EULER_AFFINE) { PUSH(L->phi, L->n_phi) PUSH(L->mu, L->n_mu) }
    PUSH(L->sigma, L->n_sigma)
#undef PUSH
  }
  return n;
}

/* mhStep, model/PMMH.scala:68-81, with init ll = -1e99 (:121), proposal perturb(delta)
 * (model/Parameters.scala:65-67: Gaussian(theta_k, sqrt(delta)) on every stored scalar),
 * symmetric transition and flat prior.  The descriptor's parameter arrays are overwritten. */
int oracle_pmmh_run(oracle_pf* pf, cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta,
                    const double* t, const double* y, const uint8_t* has, size_t T, uint64_t seed,
                    size_t n_iters, double* ll, double* theta, int32_t* accepted, double* last_state) {
  double* slots[1024];
  if (theta_slots(desc, slots, 1024) != n_theta || n_theta > 1024) return ORACLE_EINVAL;
  int d = pf->d;
  double* cur = (double*)malloc(n_theta * 8), *prop = (double*)malloc(n_theta * 8);
  double* path = (double*)malloc((T + 1) * d * 8), *cur_state = (double*)calloc(d, 8);
  memcpy(cur, theta0, n_theta * 8);
  double cur_ll = -1e99;
  int32_t acc = 0;
  double sd = sqrt(delta);
  int rc = ORACLE_OK;
  for (size_t it = 0; it < n_iters && !rc; ++it) {
    for (size_t j = 0; j < n_theta; j += 2) {                   /* proposal(s.params) */
      double z0, z1;
      cssm_normal_pair_of(seed, it, (uint32_t)(j / 2), CSSM_STREAM_HOST, 0, CSSM_LOG_TAB, &z0, &z1);
      prop[j] = cur[j] + sd * z0;
      if (j + 1 < n_theta) prop[j + 1] = cur[j + 1] + sd * z1;
    }
    for (size_t j = 0; j < n_theta; ++j) *slots[j] = prop[j];
    rc = oracle_pf_set_params(pf, desc);
    if (rc) break;
    oracle_pf_reseed(pf, cssm_derive_key(seed, (uint64_t)it + 1));
    double pll;
    int frc = oracle_pf_filter(pf, t, y, has, T, &pll, NULL, NULL, path);   /* state = pf(propParams) */
    if (frc == ORACLE_ENONFINITE) pll = -INFINITY;              /* a proposal the filter cannot weigh is rejected */
    else if (frc) { rc = frc; break; }
    double a = pll - cur_ll;                                    /* logTransition = prior = 0 */
    cssm_u32x4 b = cssm_philox_draw(seed, it, 0xffffffffu, CSSM_STREAM_HOST, 1);
    double uu = cssm_u01_open0(b.v[0], b.v[1]);
    if (cssm_log(uu) < a) {                                     /* :75 */
      cur_ll = pll; memcpy(cur, prop, n_theta * 8); memcpy(cur_state, path + T * d, d * 8); ++acc;
    }
    ll[it] = cur_ll; accepted[it] = acc;
    memcpy(theta + it * n_theta, cur, n_theta * 8);
    memcpy(last_state + it * d, cur_state, d * 8);
  }
  free(cur); free(prop); free(path); free(cur_state);
  return rc;
}

/* ------------------------------------------------------------------ contract pass-throughs */
/* So that tests can pin the numerics contract itself (Philox known answers, ulp bounds). */
double oracle_c_exp(double x) { return cssm_exp(x); }
double oracle_c_log(double x) { return cssm_log(x); }
void oracle_c_sincos2pi(double u, double* s, double* c) { cssm_sincos2pi(u, s, c); }
void oracle_c_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  cssm_u32x4 c; memcpy(c.v, ctr, 16);
  c = cssm_philox4x32_10(c, key[0], key[1]);
  memcpy(out, c.v, 16);
}
void oracle_c_philox_contract(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {   /* the contract's round count */
  cssm_u32x4 c; memcpy(c.v, ctr, 16);
  c = cssm_philox4x32(c, key[0], key[1]);
  memcpy(out, c.v, 16);
}
void oracle_c_normals(uint64_t seed, uint64_t gid, uint32_t step, uint32_t tag, uint32_t pair, double* z2) {
  cssm_normal_pair_of(seed, gid, step, tag, pair, CSSM_LOG_TAB, &z2[0], &z2[1]);
}
void oracle_c_log_unit_v(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = cssm_log_unit(x[i], CSSM_LOG_TAB); }
double oracle_c_lgamma_kp1(long long k) { return cssm_lgamma_kp1(k); }
double oracle_c_lgamma(double x) { return cssm_lgamma(x); }
uint64_t oracle_c_sys_count(double C, double u, uint64_t n) { return cssm_sys_count(C, u, n); }
double oracle_c_fix_roundtrip(double w) { return cssm_fix_to_double(cssm_fix_from_double(w)); }
double oracle_c_u(uint64_t seed, uint32_t step) {
  cssm_u32x4 b = cssm_philox_draw(seed, 0, step, CSSM_STREAM_U, 0);
  return cssm_u01(b.v[0], b.v[1]);
}
/* the accept uniform of MH iteration `it` and the key of filter run `run` (dumped as data for tests/golden/make_literal.py) */
double oracle_c_mh_u(uint64_t seed, uint64_t it) {
  cssm_u32x4 b = cssm_philox_draw(seed, it, 0xffffffffu, CSSM_STREAM_HOST, 1);
  return cssm_u01_open0(b.v[0], b.v[1]);
}
uint64_t oracle_c_derive_key(uint64_t seed, uint64_t run) { return cssm_derive_key(seed, run); }
void oracle_c_mh_normals(uint64_t seed, uint64_t it, uint64_t n_theta, double* out) {   /* the proposal's n_theta standard normals */
  for (uint64_t j = 0; j < n_theta; j += 2) {
    double z0, z1;
    cssm_normal_pair_of(seed, it, (uint32_t)(j / 2), CSSM_STREAM_HOST, 0, CSSM_LOG_TAB, &z0, &z1);
    out[j] = z0;
    if (j + 1 < n_theta) out[j + 1] = z1;
  }
}
/* the per-slot uniforms of the stratified grid / of the multinomial draws (dumped as data for tests/golden/make_literal.py) */
void oracle_c_strat_u_v(uint64_t seed, uint32_t step, uint64_t n, double* out) {
  for (uint64_t i = 0; i < n; ++i) { cssm_u32x4 b = cssm_philox_draw(seed, i, step, CSSM_STREAM_STRAT, 0); out[i] = cssm_u01(b.v[0], b.v[1]); }
}
void oracle_c_multi_u_v(uint64_t seed, uint32_t step, uint64_t n, double* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = cssm_multi_uniform(seed, step, i);
}
/* vectorised for ulp sweeps */
void oracle_c_exp_v(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = cssm_exp(x[i]); }
void oracle_c_fix_v(const double* w, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) { cssm_u128 q = cssm_fix_from_double(w[i]); out[2 * i] = q.lo; out[2 * i + 1] = q.hi; }
}
/* the d normals of particles gid0 .. gid0+n-1 of an ordinary step / the initial draw (pair streams) */
void oracle_c_paired_normals_v(uint64_t seed, uint64_t gid0, uint32_t step, uint32_t tag, int d, double* z, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint64_t gid = gid0 + i;
    const uint32_t q0 = cssm_pair_first(gid, d);
    for (int k = 0; k < d; ++k) {
      double z0, z1;
      cssm_normal_pair_of(seed, cssm_pair_stream(gid), step, tag, (q0 + (uint32_t)k) >> 1, CSSM_LOG_TAB, &z0, &z1);
      z[i * (size_t)d + k] = ((q0 + (uint32_t)k) & 1u) ? z1 : z0;
    }
  }
}
void oracle_c_log_v(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = cssm_log(x[i]); }
void oracle_c_sincos_u24_v(const uint32_t* k, double* s, double* c, size_t n) { for (size_t i = 0; i < n; ++i) cssm_sincos_u24(k[i], CSSM_LOG_TAB, &s[i], &c[i]); }
void oracle_c_sincos2pi_v(const double* u, double* s, double* c, size_t n) { for (size_t i = 0; i < n; ++i) cssm_sincos2pi(u[i], &s[i], &c[i]); }
void oracle_c_normals_v(uint64_t seed, uint64_t gid0, uint32_t step, uint32_t tag, uint32_t pair, double* z, size_t n) {
  for (size_t i = 0; i < n; ++i) cssm_normal_pair_of(seed, gid0 + i, step, tag, pair, CSSM_LOG_TAB, &z[2 * i], &z[2 * i + 1]);
}

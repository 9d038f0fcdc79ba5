// san_model.cpp -- driver of the product's host-only unit (csrc/cssm_model.cpp) for the CPU sanitizer job (oracle/Makefile,
// SAN=1): descriptor validation with well-formed and malformed descriptors, the per-observation records of every observation
// model, the LGCP sub-step table, Parameters.flattenParams order, and the PMMH accept / reject loop -- whose three calls into
// the device library (cssm_pf_set_params, cssm_pf_reseed, cssm_pf_filter; cssm_pf_dim) are stubbed here by a deterministic
// "filter" (SURVEY.md 8c, K6).  Exit status 0 and no sanitizer report = pass.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cssm_records.h"

// ---- stubs of the device library the PMMH loop calls
struct cssm_pf { int d; int calls; double last_sum; };
extern "C" int32_t cssm_pf_dim(const cssm_pf* pf) { return pf->d; }
extern "C" int cssm_pf_reseed(cssm_pf* pf, uint64_t) { pf->calls++; return CSSM_OK; }
extern "C" int cssm_pf_set_params(cssm_pf* pf, const cssm_model_desc* desc) {
  HostModel m;
  int rc = cssm_build_model(&m, desc, false);      // the real validation, on every proposal
  if (rc) return rc;
  double s = 0.0;
  for (int k = 0; k < m.d; ++k) s += m.comp[k].m0 + m.comp[k].sigma;
  pf->last_sum = s;
  return CSSM_OK;
}
extern "C" int cssm_pf_filter(cssm_pf* pf, const double*, const double*, const uint8_t*, size_t T, double* ll, double*, int32_t*, double* path) {
  *ll = -0.5 * pf->last_sum * pf->last_sum - (double)T;   // a deterministic "likelihood" of the parameters
  for (size_t i = 0; i < (T + 1) * (size_t)pf->d; ++i) path[i] = (double)i;
  return CSSM_OK;
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); ++fails; } } while (0)

static cssm_leaf_desc leaf(int sde, int dim, int f, int period, int harm, const double* m0, const double* c0, const double* mu, const double* phi,
                           const double* sg, int n) {
  cssm_leaf_desc L; memset(&L, 0, sizeof L);
  L.sde_kind = sde; L.dim = dim; L.f_kind = f; L.period = period; L.harmonics = harm;
  L.n_m0 = L.n_c0 = L.n_sigma = n; L.m0 = m0; L.c0 = c0; L.sigma = sg;
  if (mu) { L.mu = mu; L.n_mu = n; }
  if (phi) { L.phi = phi; L.n_phi = n; }
  return L;
}

int main() {
  const double m0[2] = {0.0, 0.1}, c0[2] = {0.0, -0.5}, mu[2] = {0.3, -0.2}, phi[2] = {0.2, 0.4}, sg[2] = {-1.2, -0.7};
  // ---- a composition of all four SDE kinds, seasonal leaf, every observation model
  for (int obs = CSSM_OBS_POISSON; obs <= CSSM_OBS_BETA; ++obs) {
    cssm_leaf_desc Ls[4] = {leaf(CSSM_SDE_OU, 1, CSSM_F_FIRST, 0, 0, m0, c0, mu, phi, sg, 1),
                            leaf(CSSM_SDE_OU, 2, CSSM_F_SEASONAL, 24, 1, m0, c0, mu, phi, sg, 2),
                            leaf(CSSM_SDE_GEN_BROWNIAN, 2, CSSM_F_FIRST, 0, 0, m0, c0, mu, nullptr, sg, 2),
                            leaf(CSSM_SDE_EULER_AFFINE, 1, CSSM_F_FIRST, 0, 0, m0, c0, mu, phi, sg, 1)};
    Ls[0].has_scale = 1; Ls[0].scale = -0.3;
    cssm_model_desc D; memset(&D, 0, sizeof D);
    D.n_leaves = 4; D.obs_kind = obs; D.lgcp_precision = 2; D.obs_df = 5; D.leaves = Ls;
    HostModel m;
    CHECK(cssm_build_model(&m, &D, false) == CSSM_OK);
    CHECK(m.d == 6 && m.n_leaves == 4);
    m.n_global = 1000; m.seed = 20260101;
    std::vector<StepRec> recs(6);
    double tp = 0.0;
    for (int s = 0; s < 6; ++s) {
      const double t = tp + (s == 2 ? 0.0 : 0.37 * (s + 1)), y = (obs == CSSM_OBS_BETA) ? 0.3 : (obs == CSSM_OBS_BERNOULLI ? (double)(s & 1) : 3.0 * s);
      cssm_build_rec(&m, tp, t, y, s != 4, (uint32_t)s, &recs[s]);
      CHECK(recs[s].pick < m.n_global && recs[s].u >= 0.0 && recs[s].u < 1.0);
      if (obs != CSSM_OBS_LGCP) CHECK(recs[s].ref == recs[s].ref || obs == CSSM_OBS_LGCP);
      tp = t;
    }
    std::vector<double> table;
    CHECK(cssm_build_fsub_table(&m, recs.data(), 0, recs.size(), table) == CSSM_OK);
    if (obs == CSSM_OBS_LGCP) { CHECK(m.lgcp_tdep && !table.empty()); for (const StepRec& r : recs) CHECK((size_t)r.fsub_off + (size_t)r.n_sub * m.d <= table.size()); }
    else CHECK(table.empty());
    // re-parameterise: same structure accepted, another structure refused, the model left whole
    CHECK(cssm_build_model(&m, &D, true) == CSSM_OK);
    Ls[1].period = 12;
    CHECK(cssm_build_model(&m, &D, true) == CSSM_EINVAL_DESC);
    CHECK(m.comp[1].period == 24);
    // flatten order: scale, then m0 c0 [phi mu | mu] sigma per leaf
    Ls[1].period = 24;
    size_t n = 0;
    CHECK(cssm_desc_flatten(&D, nullptr, 0, &n) == CSSM_OK);
    std::vector<double> th(n);
    CHECK(cssm_desc_flatten(&D, th.data(), n, &n) == CSSM_OK && th[0] == -0.3);
  }
  // ---- malformed descriptors are refused, never read out of bounds
  {
    HostModel m;
    cssm_model_desc D; memset(&D, 0, sizeof D);
    CHECK(cssm_build_model(&m, nullptr, false) == CSSM_EINVAL_DESC);
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    cssm_leaf_desc L = leaf(CSSM_SDE_OU, 1, CSSM_F_FIRST, 0, 0, m0, c0, nullptr, phi, sg, 1);   // OU without mu
    D.n_leaves = 1; D.leaves = &L;
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    L = leaf(CSSM_SDE_BROWNIAN, 17, CSSM_F_FIRST, 0, 0, m0, c0, nullptr, nullptr, sg, 1);     // too wide
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    L = leaf(CSSM_SDE_BROWNIAN, 3, CSSM_F_SEASONAL, 24, 1, m0, c0, nullptr, nullptr, sg, 1);   // dim != 2 harmonics
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    L = leaf(CSSM_SDE_BROWNIAN, 1, CSSM_F_FIRST, 0, 0, m0, c0, nullptr, nullptr, sg, 1);
    D.obs_kind = CSSM_OBS_GAUSSIAN;                                                            // needs a scale
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    D.obs_kind = 99;
    CHECK(cssm_build_model(&m, &D, false) == CSSM_EINVAL_DESC);
    CHECK(strlen(cssm_last_error()) > 0);
  }
  // ---- the PMMH loop over the stubbed filter: mhStep (model/PMMH.scala:68-81), first proposal always accepted (-1e99, :121)
  {
    cssm_leaf_desc L = leaf(CSSM_SDE_OU, 2, CSSM_F_FIRST, 0, 0, m0, c0, mu, phi, sg, 2);
    cssm_model_desc D; memset(&D, 0, sizeof D);
    D.n_leaves = 1; D.obs_kind = CSSM_OBS_POISSON; D.leaves = &L;
    size_t n = 0;
    CHECK(cssm_desc_flatten(&D, nullptr, 0, &n) == CSSM_OK && n == 10);
    std::vector<double> th0(n);
    cssm_desc_flatten(&D, th0.data(), n, &n);
    const size_t iters = 200, T = 7;
    std::vector<double> t(T), y(T, 1.0), ll(iters), theta(iters * n), last(iters * 2);
    for (size_t s = 0; s < T; ++s) t[s] = (double)s;
    std::vector<int32_t> acc(iters);
    cssm_pf pf = {2, 0, 0.0};
    CHECK(cssm_pmmh_run(&pf, &D, th0.data(), n, 0.05, t.data(), y.data(), nullptr, T, 7, iters, ll.data(), theta.data(), acc.data(), last.data()) == CSSM_OK);
    CHECK(acc[0] == 1 && pf.calls == (int)iters);
    for (size_t i = 1; i < iters; ++i) CHECK(acc[i] >= acc[i - 1] && acc[i] <= acc[i - 1] + 1 && (acc[i] > acc[i - 1] || ll[i] == ll[i - 1]));
    CHECK(acc[iters - 1] > 1 && acc[iters - 1] < (int)iters);
    CHECK(cssm_pmmh_run(&pf, &D, th0.data(), n - 1, 0.05, t.data(), y.data(), nullptr, T, 7, 1, ll.data(), theta.data(), acc.data(), last.data()) == CSSM_EINVAL_ARG);
  }
  printf("san_model: %s\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}

// cssm_pf.hpp -- header-only C++17 mirror of the reference's filter interface over the C ABI of cssm_pf.h.
//
// The reference is Scala; a C++ host (or a JNI layer written in C++) uses these thin wrappers.  Names and
// argument meaning follow the reference (paths relative to src/main/scala/com/github/jonnylaw/model/):
//
//   cssm::SdeParameter::brownianParameter / genBrownianParameter / ouParameter     SdeParameters.scala:192-205
//   cssm::Sde::brownianMotion / genBrownianMotion / ouProcess                        Sde.scala:181-202
//   cssm::Model::poisson / linear / seasonal / lgcp / negativeBinomial / ...         Model.scala:44-95
//   a | b  (the reference's  a |+| b)                                                Model.scala:97-136
//   cssm::Filter(mod, n).llFilter / filter / initialiseState / stepFilter            ParticleFilter.scala:96-167
//
// Errors: a non-zero status of the C ABI becomes cssm::Error carrying cssm_last_error() -- the reference
// throws from inside stepFilter (Sde.scala:214, Model.scala:150).
#ifndef CSSM_PF_HPP
#define CSSM_PF_HPP

#include <cmath>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "cssm_pf.h"

namespace cssm {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("cssm error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != CSSM_OK) throw Error(rc, cssm_last_error());
}

using Vec = std::vector<double>;

// STORED (unconstrained) SDE parameters, SdeParameters.scala:50-160
struct SdeParam {
  int kind = CSSM_SDE_BROWNIAN;
  Vec m0, c0, mu, phi, sigma;
};

struct SdeParameter {
  static Vec logv(Vec v) { for (auto& x : v) x = std::log(x); return v; }
  static double logistic(double x) { return 1.0 / (1.0 + std::exp(-x)); }                    // SdeParameters.scala:214-216
  static SdeParam brownianParameter(Vec m0, Vec c0, Vec sigma) {                               // :197-200
    return {CSSM_SDE_BROWNIAN, std::move(m0), logv(std::move(c0)), {}, {}, logv(std::move(sigma))};
  }
  static SdeParam genBrownianParameter(Vec m0, Vec c0, Vec mu, Vec sigma) {                    // :192-195
    return {CSSM_SDE_GEN_BROWNIAN, std::move(m0), logv(std::move(c0)), std::move(mu), {}, logv(std::move(sigma))};
  }
  static SdeParam ouParameter(Vec m0, Vec c0, Vec phi, Vec mu, Vec sigma) {                    // :202-205 (phi.map(logistic), sic)
    for (auto& p : phi) p = logistic(p);
    return {CSSM_SDE_OU, std::move(m0), logv(std::move(c0)), std::move(mu), std::move(phi), logv(std::move(sigma))};
  }
};

struct UnparamSde { int kind; int dimension; };
struct Sde {
  static UnparamSde brownianMotion(int d) { return {CSSM_SDE_BROWNIAN, d}; }
  static UnparamSde genBrownianMotion(int d) { return {CSSM_SDE_GEN_BROWNIAN, d}; }
  static UnparamSde ouProcess(int d) { return {CSSM_SDE_OU, d}; }
};

struct ParamNode {                                    // Parameters.scala:14
  std::optional<double> scale;
  SdeParam sdeParam;
};
using Parameters = std::vector<ParamNode>;             // leaves in Tree.flatten order
inline Parameters operator|(Parameters a, const Parameters& b) { a.insert(a.end(), b.begin(), b.end()); return a; }

struct LeafSpec { int obs_kind; int f_kind; UnparamSde sde; int period = 0, harmonics = 0, df = 0; };

// ReaderT[Try, Parameters, Model]: the unparameterised (possibly composed) model
struct UnparamModel {
  std::vector<LeafSpec> leaves;
  friend UnparamModel operator|(UnparamModel a, const UnparamModel& b) {
    a.leaves.insert(a.leaves.end(), b.leaves.begin(), b.leaves.end());
    return a;
  }
};

// A parameterised model: owns the arrays behind a cssm_model_desc
class ParamModel {
 public:
  ParamModel(const UnparamModel& m, Parameters p, int lgcp_precision = 0) : params_(std::move(p)) {
    if (m.leaves.size() != params_.size()) throw std::invalid_argument("parameter tree shape does not match the composed model");
    leaves_.resize(m.leaves.size());
    for (size_t i = 0; i < m.leaves.size(); ++i) {
      const LeafSpec& s = m.leaves[i];
      const ParamNode& n = params_[i];
      if (n.sdeParam.kind != s.sde.kind) throw std::invalid_argument("Incorrect parameters supplied to the SDE of leaf " + std::to_string(i));
      cssm_leaf_desc& L = leaves_[i];
      L = cssm_leaf_desc{};
      L.sde_kind = s.sde.kind; L.dim = s.sde.dimension; L.f_kind = s.f_kind; L.period = s.period; L.harmonics = s.harmonics;
      L.has_scale = n.scale ? 1 : 0; L.scale = n.scale.value_or(0.0);
      auto set = [](const Vec& v, const double*& p, int32_t& cnt) { p = v.empty() ? nullptr : v.data(); cnt = (int32_t)v.size(); };
      set(n.sdeParam.m0, L.m0, L.n_m0); set(n.sdeParam.c0, L.c0, L.n_c0); set(n.sdeParam.mu, L.mu, L.n_mu);
      set(n.sdeParam.phi, L.phi, L.n_phi); set(n.sdeParam.sigma, L.sigma, L.n_sigma);
    }
    desc_.n_leaves = (int32_t)leaves_.size();
    desc_.obs_kind = m.leaves.front().obs_kind;          // only the leftmost model observes, Model.scala:118-132
    desc_.lgcp_precision = lgcp_precision;
    desc_.obs_df = m.leaves.front().df;
    desc_.leaves = leaves_.data();
  }
  ParamModel(const ParamModel&) = delete;
  ParamModel& operator=(const ParamModel&) = delete;
  const cssm_model_desc* desc() const { return &desc_; }
  int dimension() const { int d = 0; for (auto& l : leaves_) d += l.dim; return d; }

 private:
  Parameters params_;
  std::vector<cssm_leaf_desc> leaves_;
  cssm_model_desc desc_{};
};

struct Model {                                          // Model.scala:44-95
  static UnparamModel leaf(int obs, int f, UnparamSde s, int period = 0, int harmonics = 0, int df = 0) {
    return UnparamModel{{LeafSpec{obs, f, s, period, harmonics, df}}};
  }
  static UnparamModel poisson(UnparamSde s) { return leaf(CSSM_OBS_POISSON, CSSM_F_FIRST, s); }
  static UnparamModel linear(UnparamSde s) { return leaf(CSSM_OBS_GAUSSIAN, CSSM_F_FIRST, s); }
  static UnparamModel seasonal(int period, int harmonics, UnparamSde s) { return leaf(CSSM_OBS_GAUSSIAN, CSSM_F_SEASONAL, s, period, harmonics); }
  static UnparamModel lgcp(UnparamSde s) { return leaf(CSSM_OBS_LGCP, CSSM_F_FIRST, s); }
  static UnparamModel negativeBinomial(UnparamSde s) { return leaf(CSSM_OBS_NEGBIN, CSSM_F_FIRST, s); }
  static UnparamModel zeroInflatedPoisson(UnparamSde s) { return leaf(CSSM_OBS_ZIP, CSSM_F_FIRST, s); }
  static UnparamModel bernoulli(UnparamSde s) { return leaf(CSSM_OBS_BERNOULLI, CSSM_F_FIRST, s); }
  static UnparamModel studentsT(UnparamSde s, int df) { return leaf(CSSM_OBS_STUDENT_T, CSSM_F_FIRST, s, 0, 0, df); }
  static UnparamModel beta(UnparamSde s) { return leaf(CSSM_OBS_BETA, CSSM_F_FIRST, s); }
};

struct TimedObservation { double t; std::optional<double> observation; };   // Data.scala:18-21
using Data = TimedObservation;

struct PfState { double t; std::optional<double> observation; double ll; int ess; };   // ParticleFilter.scala:32-37 (cloud stays on the device)
struct StateSpace { double time; Vec state; };

// object Resampling (model/Resampling.scala): the `Resample[A]` of package.scala:23 on host vectors -- the ancestor indices
// come from the GPU (cssm_resample), the gather of the caller's `A`s happens here.  `weights` are the unnormalised
// w1 = exp(w - max) the reference passes (ParticleFilter.scala:125-126).  The reference draws u / the per-slot uniforms
// from unseeded global generators (:66, :83, :93); here the caller passes u, or (seed, step) of the contract's streams.
struct Resampling {
  static std::vector<uint32_t> ancestors(int kind, const Vec& weights, double u = 0.0, uint64_t seed = 0, uint32_t step = 0, int device = 0) {
    std::vector<uint32_t> anc(weights.size());
    check(cssm_resample(kind, weights.data(), weights.size(), u, seed, step, anc.data(), device));
    return anc;
  }
  template <class A>
  static std::vector<A> gather(const std::vector<A>& particles, const std::vector<uint32_t>& anc) {
    std::vector<A> out; out.reserve(anc.size());
    for (uint32_t a : anc) out.push_back(particles[a]);
    return out;
  }
  template <class A>
  static std::vector<A> systematicResampling(const std::vector<A>& particles, const Vec& weights, double u, int device = 0) {   // :63-72
    if (particles.size() != weights.size()) throw Error(CSSM_EINVAL_ARG, "particles and weights differ in length");
    return gather(particles, ancestors(CSSM_RESAMPLE_SYSTEMATIC, weights, u, 0, 0, device));
  }
  template <class A>
  static std::vector<A> stratifiedResampling(const std::vector<A>& particles, const Vec& weights, uint64_t seed, uint32_t step = 0, int device = 0) {   // :78-86
    if (particles.size() != weights.size()) throw Error(CSSM_EINVAL_ARG, "particles and weights differ in length");
    return gather(particles, ancestors(CSSM_RESAMPLE_STRATIFIED, weights, 0.0, seed, step, device));
  }
  template <class A>
  static std::vector<A> multinomialResampling(const std::vector<A>& particles, const Vec& weights, uint64_t seed, uint32_t step = 0, int device = 0) {   // :92-96
    if (particles.size() != weights.size()) throw Error(CSSM_EINVAL_ARG, "particles and weights differ in length");
    return gather(particles, ancestors(CSSM_RESAMPLE_MULTINOMIAL, weights, 0.0, seed, step, device));
  }
};

// Filter(mod, resample) with resample = Resampling.systematicResampling (or CSSM_RESAMPLE_* through `resampler`)
class Filter {
 public:
  Filter(const ParamModel& mod, uint64_t particles, uint64_t seed = 20260101, int device = 0, int resampler = CSSM_RESAMPLE_SYSTEMATIC)
      : d_(mod.dimension()), n_(particles) {
    check(cssm_pf_create(mod.desc(), particles, seed, device, &h_));
    if (resampler != CSSM_RESAMPLE_SYSTEMATIC) check(cssm_pf_set_option(h_, CSSM_OPT_RESAMPLER, resampler));
  }
  ~Filter() { cssm_pf_destroy(h_); }
  Filter(const Filter&) = delete;
  Filter& operator=(const Filter&) = delete;

  PfState initialiseState(double t0) {                                           // ParticleFilter.scala:105-108
    check(cssm_pf_init(h_, t0));
    return {t0, std::nullopt, 0.0, (int)n_};
  }
  PfState stepFilter(const PfState&, const Data& y) {                            // :116-132
    double ll; int32_t ess;
    check(cssm_pf_step(h_, y.t, y.observation.value_or(0.0), y.observation ? 1 : 0, &ll, &ess));
    return {y.t, y.observation, ll, ess};
  }
  double llFilter(const std::vector<Data>& data) {                               // :137-140
    Vec t, y; std::vector<uint8_t> has; split(data, t, y, has);
    double ll;
    check(cssm_pf_ll_filter(h_, t.data(), y.data(), has.data(), t.size(), &ll, nullptr, nullptr));
    return ll;
  }
  std::pair<double, std::vector<StateSpace>> filter(const std::vector<Data>& data) {   // :152-158
    Vec t, y; std::vector<uint8_t> has; split(data, t, y, has);
    Vec path((t.size() + 1) * (size_t)d_);
    double ll;
    check(cssm_pf_filter(h_, t.data(), y.data(), has.data(), t.size(), &ll, nullptr, nullptr, path.data()));
    double t0 = t[0];
    for (double v : t) t0 = std::min(t0, v);
    std::vector<StateSpace> out;
    for (size_t s = 0; s <= t.size(); ++s) out.push_back({s == 0 ? t0 : t[s - 1], Vec(path.begin() + s * d_, path.begin() + (s + 1) * d_)});
    return {ll, out};
  }
  Vec particles() { Vec out((size_t)d_ * n_); check(cssm_pf_get_particles(h_, out.data())); return out; }   // SoA [d][N]
  void setParams(const ParamModel& mod, uint64_t seed) { check(cssm_pf_set_params(h_, mod.desc())); check(cssm_pf_reseed(h_, seed)); }
  cssm_pf* handle() { return h_; }

 private:
  static void split(const std::vector<Data>& data, Vec& t, Vec& y, std::vector<uint8_t>& has) {
    for (auto& d : data) { t.push_back(d.t); y.push_back(d.observation.value_or(0.0)); has.push_back(d.observation ? 1 : 0); }
  }
  cssm_pf* h_ = nullptr;
  int d_;
  uint64_t n_;
};

}  // namespace cssm
#endif

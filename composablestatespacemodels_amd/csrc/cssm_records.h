// cssm_records.h -- the records that cross from the host model code to the kernels: per-model constants (ModelK) and the
// per-observation record (StepRec).  Plain C++ (no HIP): cssm_model.cpp -- descriptor validation, build_rec, the PMMH loop --
// compiles with any host compiler (and under sanitizers: oracle/Makefile, target `san`), the kernels include it through
// cssm_device.hip.h.
#pragma once

#if !defined(__HIPCC_RTC__)
#include <stdint.h>

#include <vector>
#endif

#include "../../include/cssm_numerics.h"
#include "../../include/cssm_pf.h"

#if defined(__HIPCC__)
#define CSSM_HDM __host__ __device__ __forceinline__
#else
#define CSSM_HDM inline
#endif

// f-map modes per component (host-built from the leaf list; oracle: gamma_of)
#define FM_SKIP 0
#define FM_START 1 /* first used component of a leaf: acc = c*x  */
#define FM_ADD 2   /* acc += c*x                                 */

// Per-model constants, passed BY VALUE (kernarg -> SGPRs; every branch on them is wave-uniform).
// One byte per latent component: bits 0-1 CSSM_SDE_*, bits 2-3 FM_*, bit 4 closes its leaf, bit 5 leaf is
// the leftmost one.  Packed four to a word so that the whole struct costs 6 SGPRs (66 unpacked made the
// kernel spill scalars through v_writelane/v_readlane).
struct ModelK {
  int32_t d;
  int32_t obs_kind;
  uint32_t comp[CSSM_MAX_DIM / 4];
  CSSM_HDM uint32_t byte(int k) const { return (comp[k >> 2] >> ((k & 3) * 8)) & 0xffu; }
  CSSM_HDM int kind(int k) const { return (int)(byte(k) & 3u); }
  CSSM_HDM int fmode(int k) const { return (int)((byte(k) >> 2) & 3u); }
  CSSM_HDM bool leaf_end(int k) const { return (byte(k) >> 4) & 1u; }
  CSSM_HDM bool first_leaf(int k) const { return (byte(k) >> 5) & 1u; }
};

// Per-observation record, built on the host (everything that depends only on (t, y)): build_rec, cssm_model.cpp.
struct StepRec {
  double y;       // count models: (double)trunc(y); otherwise y
  double c[4];    // per-observation constants of the density (see build_rec / logdens)
  double cdf;     // Student-t: degrees of freedom as double
  double u;       // the one uniform of systematic resampling
  double dt;      // time increment (LGCP: the sub-step delta)
  double ref;     // reference level of the observation (cssm_ref_level; NaN: always rescale by the max)
  int32_t has_obs;
  int32_t n_sub;  // LGCP sub-steps (0: dt == 0, weight 0, state kept)
  uint32_t pick;  // sampleOne index for `filter`
  uint32_t step;  // observation index (Philox counter word 2)
  double coef[CSSM_MAX_DIM][4]; // transition coefficients per component
  double fco[CSSM_MAX_DIM];     // f coefficients c_k(t)
  double t_obs;                 // the observation's time (LGCP with a time-dependent f: the sub-step clock starts here)
  uint32_t fsub_off;            // LGCP with a time-dependent f: where this observation's n_sub x d coefficients c_k(tau_s) start
                                //   in the handle's sub-step table (doubles)
  uint32_t predict;             // 1 (LGCP): `ref` is not a function of the observation -- it is the level PREDICTED from the max of the weighted
                                //   observation before (cssm_ref_predict), which the kernel that publishes that observation's scalars writes into
                                //   the NEXT record (rec[1].ref) and into Scalars::next_ref; the host uploads NaN and chains the first record of a
                                //   call from Scalars::next_ref on the stream (cssm_upload_recs).  0: ref = cssm_ref_level of the observation.
};

// One latent component of the composed model, constraint transforms applied (model/Sde.scala:70-73,99-102,133-137).
struct Comp {
  int kind, leaf, idx, f_kind, period;
  double m0, c0, mu, phi, sigma;
};

// The model as the host sees it: what descriptor validation produces and build_rec consumes.  cssm_pf (cssm_internal.h)
// derives from it; nothing here touches a device.
struct HostModel {
  int d = 0, n_leaves = 0, obs_kind = 0, precision = 0, obs_df = 0;
  double scale_sd = 1.0;       // exp(scale): Gaussian sd, NegBin size, Student-t v
  double scale_raw = 0.0;      // ZIP: the stored scale v
  Comp comp[CSSM_MAX_DIM];
  ModelK mk;
  bool lgcp_tdep = false;      // LGCP whose f depends on time (a seasonal leaf): f is evaluated at every sub-step time
  uint64_t n_global = 0, seed = 0;   // (build_rec: sampleOne's index, the Philox key of the resampling uniform)
};

// ---- cssm_model.cpp (host only) ----
// one PMMH chain's state (model/PMMH.scala:68-81): shared by the sequential and the batched driver
struct cssm_pmmh_chain;
int cssm_pmmh_chain_create(const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, uint64_t seed, int d, cssm_pmmh_chain** out);
void cssm_pmmh_chain_destroy(cssm_pmmh_chain* c);
const cssm_model_desc* cssm_pmmh_chain_propose(cssm_pmmh_chain* c, size_t it, uint64_t* key_out);
void cssm_pmmh_chain_decide(cssm_pmmh_chain* c, size_t it, double pll, const double* last_path_row, double* ll_out, double* theta_out, int32_t* acc_out,
                            double* state_out);
// (the speculative driver, cssm_pmmh_run_speculative: proposals from a given parameter vector, the decision on a given proposal)
void cssm_pmmh_chain_set_current(cssm_pmmh_chain* c, const double* theta);
void cssm_pmmh_chain_set_proposal(cssm_pmmh_chain* c, const double* theta);
const double* cssm_pmmh_chain_current(const cssm_pmmh_chain* c);
const double* cssm_pmmh_chain_proposal(const cssm_pmmh_chain* c);
int32_t cssm_pmmh_chain_accepted(const cssm_pmmh_chain* c);
int cssm_fail(int code, const char* fmt, ...);   // sets the thread-local message of cssm_last_error(), returns code
// Validate and translate a descriptor; `update`: re-parameterise -- the STRUCTURE must be the one `m` already has.
int cssm_build_model(HostModel* m, const cssm_model_desc* desc, bool update);
// Everything of one observation that does not depend on the particle (records for the kernels).
void cssm_build_rec(const HostModel* m, double t_prev, double t, double y, int has_obs, uint32_t step, StepRec* r);
// LGCP with a time-dependent f (FilterLgcp.calcWeight evaluates f at every simulated time, model/ParticleFilter.scala:193-205):
// append the n_sub x d coefficient rows of records recs[first .. first + count) to `table` and set their fsub_off;
// CSSM_ENOMEM if the table would exceed 1 GiB.  No-op for every other model.
#if !defined(__HIPCC_RTC__)
int cssm_build_fsub_table(const HostModel* m, StepRec* recs, size_t first, size_t count, std::vector<double>& table);
#endif

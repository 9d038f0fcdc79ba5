/*
 * cssm_pf.h -- C ABI of libcssm_pf, the MI355X (gfx950) bootstrap particle filter.
 *
 * The reference (jonnylaw/ComposableStateSpaceModels, Scala/JVM) has no FFI; this header is
 * the FFI its `ParticleFilter` seam would bind.  Every entry point
 * cites the reference symbol it replaces as  model/<File>.scala:<lines>  (paths relative to
 * src/main/scala/com/github/jonnylaw/ of the reference).  The JNI/Scala binding a maintainer
 * would add is shown in INTEGRATION.md and shipped as source under jvm/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, all floating point is IEEE fp64.
 *   - Return value 0 (CSSM_OK) or a negative CSSM_E* code; cssm_last_error() returns a
 *     thread-local message for the last failing call on this thread.  The reference throws
 *     from inside stepFilter (model/Sde.scala:214, model/Model.scala:150); the JNI glue turns
 *     a non-zero code into a RuntimeException.
 *   - The handle owns all device memory and its HIP stream; the caller owns every
 *     host buffer passed in or out.  Pointers documented as "host" must be host memory,
 *     pointers documented as "device" must be device memory of the handle's GPU.
 *   - A handle is not re-entrant; distinct handles share nothing mutable and may be driven
 *     from different threads (two PMMH chains: examples/DetermineParameters.scala:68-69).
 *     Every entry point selects the handle's device itself, so calls may arrive on any thread
 *     (Akka dispatcher threads: model/ParticleFilter.scala:163-166).
 *   - Random numbers and reductions follow include/cssm_numerics.h (the numerics contract).
 *   - Stated fp64 tolerance to the reference's literal arithmetic (sequential fp64 sums and cumulative weights, platform
 *     libm, rescaling by the max -- model/ParticleFilter.scala:124-128, model/Resampling.scala:21-24,57) under the same
 *     variates:  |ll - ll_literal| <= 1e-9 * T  for a series of T observations (measured: 1e-13), and at most a fraction
 *     1e-4 of the ancestor indices of the first weighted observation differ (measured: none).  The reference's TreeMap
 *     keeps the LAST particle of equal cumulative weights (:57); the library keeps the first, which owns the mass
 *     (deviation D3): with that quirk switched on in the oracle, 0.4 % of the slots at N = 2^16 go to a
 *     particle of negligible weight instead, after which the two runs agree as two realisations of one estimator do.
 *     Against the CPU restatement that shares the contract (oracle/), everything is bit-identical.
 *     tests/test_gpu_parity.py asserts all of it on the GPU.
 */
#ifndef CSSM_PF_H
#define CSSM_PF_H

#if !defined(__HIPCC_RTC__)
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------------------- */
#define CSSM_OK 0
#define CSSM_EINVAL_DESC (-1) /* malformed model descriptor (reference: Failure(...) at model/Model.scala:44-47) */
#define CSSM_EHIP (-2)        /* a HIP runtime call failed / no gfx950 device */
#define CSSM_ESHARD (-3)      /* inconsistent sharding arguments */
#define CSSM_ENOMEM (-4)
#define CSSM_ENONFINITE (-5)  /* NaN log-weight or all weights zero (reference: breeze `require` throws) */
#define CSSM_EINVAL_ARG (-6)
#define CSSM_ESTATE (-7)      /* call out of sequence (e.g. step before init) */
#define CSSM_ERCCL (-8)       /* librccl.so could not be loaded, or an RCCL call failed */

/* ---- model descriptor ------------------------------------------------------------------- */
/* Latent SDE of one leaf: model/Sde.scala:98-124 (Brownian), :69-96 (GenBrownian),
 * :129-163 (OU), :23-43 (trait default Euler-Maruyama, here with affine drift a + b*x and
 * constant diagonal diffusion g). */
#define CSSM_SDE_BROWNIAN 0
#define CSSM_SDE_GEN_BROWNIAN 1
#define CSSM_SDE_OU 2
#define CSSM_SDE_EULER_AFFINE 3

/* Linear map f of one leaf: "first component" model/Model.scala:271 (also :153,184,250,296,
 * 328,347,366) or seasonal model/Model.scala:217-225. */
#define CSSM_F_FIRST 0
#define CSSM_F_SEASONAL 1

/* Observation model of the LEFTMOST leaf (model/Model.scala:118-132):
 * Poisson model/Model.scala:266-274; Gaussian (LinearModel :241-259, SeasonalModel :227-233);
 * LGCP model/Model.scala:363-369 with the FilterLgcp step model/ParticleFilter.scala:169-227. */
#define CSSM_OBS_POISSON 0
#define CSSM_OBS_GAUSSIAN 1
#define CSSM_OBS_LGCP 2
/* "next" rows of SURVEY.md 8f-1, all with f = first component of the leftmost leaf: */
#define CSSM_OBS_NEGBIN 3     /* NegativeBinomialModel  model/Model.scala:168-196, size = exp(scale)          */
#define CSSM_OBS_ZIP 4        /* ZeroInflatedPoisson    model/Model.scala:281-309, p = logistic(scale)        */
#define CSSM_OBS_BERNOULLI 5  /* BernoulliModel         model/Model.scala:315-337 (+-6 clamp, -1e99 floor)    */
#define CSSM_OBS_STUDENT_T 6  /* StudentsTModel         model/Model.scala:144-161, v = exp(scale), obs_df     */
#define CSSM_OBS_BETA 7       /* BetaModel              model/Model.scala:339-353, Beta(exp(-gamma), 1)       */

#define CSSM_MAX_DIM 16     /* total latent dimension d (sum over leaves) */
#define CSSM_MAX_LEAVES 16

/*
 * One leaf of the composed model, left-to-right in Tree.flatten order (model/Tree.scala:49-53).
 * Parameter arrays hold the STORED (unconstrained) values exactly as ParamNode/SdeParameter
 * hold them (model/SdeParameters.scala:176-205) -- log c0, log sigma, and for OU the value
 * the reference later passes through `logistic` (model/Sde.scala:136).  The library applies
 * exp/logistic itself (cssm_exp) and repeats each array cyclically to `dim`
 * (Sde.buildParamRepeat, model/Sde.scala:177-179).  For CSSM_SDE_EULER_AFFINE mu = a,
 * phi = b, sigma = g are taken as they are (a user-defined Sde has no stored transform);
 * c0 is still a log-variance.
 */
typedef struct cssm_leaf_desc {
  int32_t sde_kind;
  int32_t dim;
  int32_t f_kind;
  int32_t period;     /* CSSM_F_SEASONAL: model/Model.scala:205 */
  int32_t harmonics;  /* CSSM_F_SEASONAL: dim must equal 2*harmonics */
  int32_t has_scale;  /* ParamNode.scale is Some(_) (model/Parameters.scala:14) */
  double scale;       /* stored scale; Gaussian uses sd = exp(scale) (model/Model.scala:244) */
  int32_t n_m0, n_c0, n_mu, n_phi, n_sigma;
  int32_t reserved;
  const double* m0;
  const double* c0;
  const double* mu;    /* GenBrownian, OU, EulerAffine(a) */
  const double* phi;   /* OU, EulerAffine(b) */
  const double* sigma; /* all kinds */
} cssm_leaf_desc;

typedef struct cssm_model_desc {
  int32_t n_leaves;
  int32_t obs_kind;
  int32_t lgcp_precision; /* FilterLgcp.precision (model/ParticleFilter.scala:172) */
  int32_t obs_df;         /* CSSM_OBS_STUDENT_T: degrees of freedom (`df: Int`, model/Model.scala:144) */
  const cssm_leaf_desc* leaves;
} cssm_model_desc;

typedef struct cssm_pf cssm_pf; /* opaque handle */

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Build a filter for `n_particles` particles on HIP device `device`.  Replaces the
 * construction `Filter(mod, resample)` / `FilterLgcp(mod, resample, precision)`
 * (model/ParticleFilter.scala:233-235, :169-172) with resample = systematicResampling. */
int cssm_pf_create(const cssm_model_desc* desc, uint64_t n_particles, uint64_t seed, int device,
                   cssm_pf** out);

/* Same, for one shard of a filter whose N_global particles are split over `world` GPUs (one
 * process per GPU).  This rank owns global particles and resampling slots
 * [first, first + n_local).  Random variates are keyed by GLOBAL particle id, so results do
 * not depend on `world`.  The collectives between the stages are the caller's (torch.distributed
 * over RCCL): see the cssm_pf_shard_* stage calls below. */
int cssm_pf_create_shard(const cssm_model_desc* desc, uint64_t n_global, uint64_t first,
                         uint64_t n_local, uint64_t seed, int device, void* hip_stream,
                         cssm_pf** out);

void cssm_pf_destroy(cssm_pf* pf);

/* New parameters for the same model structure, without reallocating: one call per PMMH
 * proposal (model/PMMH.scala:71, `pf(propParams)`). */
int cssm_pf_set_params(cssm_pf* pf, const cssm_model_desc* desc);

/* New Philox key (a fresh filter run must not reuse variates). */
int cssm_pf_reseed(cssm_pf* pf, uint64_t seed);
/* The key a driver should hand to cssm_pf_reseed for its `run`-th filter run under one user seed: a PRF of (seed, run)
 * (include/cssm_numerics.h, cssm_derive_key), never seed + run -- chains seeded s and s + 1 would otherwise replay each
 * other's filter randomness one iteration apart.  cssm_pmmh_run uses run = iteration + 1. */
uint64_t cssm_pf_run_key(uint64_t seed, uint64_t run);

/* ---- streaming mode: one native call per observation -------------------------------------- */

/* initialiseState (model/ParticleFilter.scala:105-108): N draws m0 + sqrt(c0) z; ll = 0,
 * ess = N, t = t0, step counter = 0. */
int cssm_pf_init(cssm_pf* pf, double t0);

/* FilterInit.initialiseState (model/ParticleFilter.scala:257-260): replicate one state
 * (host pointer, d doubles in flatten order) N times. */
int cssm_pf_init_from(cssm_pf* pf, double t0, const double* state_d);

/* stepFilter (model/ParticleFilter.scala:116-132; LGCP: :210-226).  `has_obs` = 0 is the
 * `None` branch (:121): propagate only, ll and ess unchanged.  ll_out = accumulated
 * log-likelihood, ess_out = floor(1 / sum(normalised w^2)) (:431-434). */
int cssm_pf_step(cssm_pf* pf, double t, double obs, int has_obs, double* ll_out, int32_t* ess_out);

/* stepFilter split at the resampler, for a `Resample[A]` the library does not have natively (any host function:
 * model/package.scala:23; Resampling.indentity, model/Resampling.scala:29).  cssm_pf_propagate does :117-124 (propagate the
 * cloud to t, weigh it; `has_obs` = 0: propagate only); the host reads the proposed cloud and the log-weights
 * (cssm_pf_get_proposed, cssm_pf_get_logw), forms w1 = exp(w - max), applies its resampler, and hands the resampled cloud
 * (SoA, d x N, host) back with cssm_pf_adopt together with the new ll and ess (:126-130).  A parity path: the cloud crosses
 * PCIe twice per observation.  After cssm_pf_propagate without cssm_pf_adopt the proposed cloud is the current one. */
int cssm_pf_propagate(cssm_pf* pf, double t, double obs, int has_obs);
int cssm_pf_adopt(cssm_pf* pf, const double* state_dN, double ll, int32_t ess);

/* ---- batch mode: one native call for all T observations ----------------------------------- */

/* llFilter (model/ParticleFilter.scala:137-140): t0 = min t, init, fold stepFilter over the
 * data in the order given.  Optional outputs (may be NULL): ll_t[T] running log-likelihood
 * after each datum, ess_t[T].  No host synchronisation happens inside the loop. */
int cssm_pf_ll_filter(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs,
                      size_t T, double* ll_out, double* ll_t, int32_t* ess_t);
/* The same for T MORE observations of a filter that is already running (initialised by cssm_pf_init / _init_from, or left
 * behind by a batch run or by streaming steps): no initial cloud is drawn, the first time increment is taken from the
 * handle's clock, and observation s of this call is observation (steps so far + s) of the filter -- what Flow.scan(init)
 * (stepFilter) does with the next T elements of the stream (model/ParticleFilter.scala:163-166), without a host round trip
 * per element.  *ll_out = the log-likelihood accumulated since the filter was initialised; ll_t / ess_t: this call's
 * observations.  cssm_pf_ll_filter(t[0..a)) followed by cssm_pf_ll_filter_more(t[a..T)) leaves the same bits as
 * cssm_pf_ll_filter(t[0..T)). */
int cssm_pf_ll_filter_more(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                           double* ll_out, double* ll_t, int32_t* ess_t);

/* filter (model/ParticleFilter.scala:152-158): as llFilter, and path[(T+1)*d] receives one
 * uniformly picked particle of the initial cloud and of the cloud after every datum
 * (Resampling.sampleOne, model/Resampling.scala:151-154), row s = time index, d doubles. */
int cssm_pf_filter(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                   double* ll_out, double* ll_t, int32_t* ess_t, double* path);

/* Device time of the last cssm_pf_ll_filter / cssm_pf_filter loop (HIP events on the
 * handle's stream around the T steps, init excluded), in milliseconds.  The events are recorded only
 * with CSSM_OPT_LOOP_EVENTS = 1 (below; environment CSSM_LOOP_EVENTS=1 sets that default): CSSM_ESTATE otherwise. */
int cssm_pf_last_loop_ms(cssm_pf* pf, float* ms_out);

/* Device time of the last batch call (cssm_pf_ll_filter_more, a sharded cssm_pf_shard_continue + series + status) WITHOUT event packets:
 * the kernel that brings the call's first record stamps the GPU's constant 100 MHz clock (s_memrealtime) at its first instruction, the
 * call's closing kernel (k_finish) stamps it again and hands both to the host with the results.  *us_out = first instruction of the
 * call's first kernel -> the closing kernel's copy of the results, in microseconds (resolution 10 ns).  What bench.py reports per
 * timed leg (`device_ms_each`) beside the host's wall time.  CSSM_ESTATE if the last call left no pair of stamps (a call that drew
 * a new cloud: the scalars are reset behind its first kernel). */
int cssm_pf_last_device_us(cssm_pf* pf, double* us_out);

/* 1 if nothing is queued or running on the handle's stream (hipStreamQuery: every launch of the calls so far has completed as the runtime
 * sees it), 0 if work is still pending, negative on a HIP error.  The batch drivers return on their closing kernel's completion word;
 * a caller that wants the runtime's own confirmation (bench.py does, right behind its timed region) asks here -- no wait, no packet. */
int cssm_pf_stream_idle(cssm_pf* pf);

/* Debug/verification options.  CSSM_OPT_EXACT_OFFSPRING = 1 makes the offspring kernel evaluate the
 * contract's exact predicate for every particle instead of only where its fp64 position estimate is
 * within the error band of a slot boundary; results are identical by construction (tests compare).
 * Value 2 (systematic resampling): every third particle only -- threads then hand over single particles, as the
 * fast path does when one of them is close to a boundary. */
#define CSSM_OPT_EXACT_OFFSPRING 1
/* CSSM_OPT_RESAMPLER selects the `Resample[A]` the filter was constructed with (model/ParticleFilter.scala:
 * 233-235): systematic (model/Resampling.scala:63-72, default), stratified (:78-86) or multinomial (:92-96).
 * The reference's residualResampling (:130-146) indexes `Vector.range(1, m)` with draws from [0, n) and cannot
 * run as written; it is not offered.  Sharded handles support systematic and stratified resampling (the grid points of both are
 * keyed by GLOBAL slot and ordered, so the ancestors of a rank's slots are its own particles plus its neighbours' boundary blocks);
 * a multinomial resampler draws every slot's ancestor from the whole cloud and stays single-GPU. */
#define CSSM_OPT_RESAMPLER 2
#define CSSM_RESAMPLE_SYSTEMATIC 0
#define CSSM_RESAMPLE_STRATIFIED 1
#define CSSM_RESAMPLE_MULTINOMIAL 2
/* CSSM_OPT_FUSED_SUMS (default 1; a verification switch like CSSM_OPT_WHOLE_TILES -- results are identical, both settings apply
 * cssm_ref_choose): k_propagate also forms the fixed-point sums of exp(w - c) (c = the observation's reference level,
 * cssm_numerics.h) -- the log-sum-exp normalisation inside the fused kernel: 2 kernels per observation.  When the max rules c
 * out (an outlying observation) the batch drivers put the series on hold at that observation, redo its sums relative to the
 * max and carry on; streaming cssm_pf_step does the same within the call.  0 = the sums are always a pass of their own after
 * the max is known (k_tile_sums, 3 kernels per observation) -- the path LGCP, the multinomial resampler and a redone observation
 * run anyway; measured slower at every size (20.0 vs 18.7 us at N = 100 000, 35.4 vs 34.0 at 2^20, 356 vs 336 at 2^24). */
#define CSSM_OPT_FUSED_SUMS 3
/* CSSM_OPT_WHOLE_TILES (default 0): a verification switch over the launch geometry of the propagate kernel (results are
 * bit-identical in every setting).  Automatic (0): clouds below 2^20 particles run ONE tile of the kernel per block (512
 * particles for d <= 8, 256 for d >= 9: the single-tile kernel requests everything position-dependent in its first round of
 * loads and draws the normals while the gathered rows travel; k_offspring totals one pair of sums per block); from 2^20 on a
 * block owns a whole unit of 1024 * k particles and runs (1) the same kernel tile after tile while a unit has at most 8
 * tiles, else (2) the software-pipelined kernel (d <= 3) or (3) one tile per block again with k_reduce_units folding the
 * blocks' sums into <= 1024 unit sums (d >= 4).  1 / 2 / 3 force that geometry at every size: the kernels of the large
 * clouds can then be checked against the oracle at sizes the oracle finishes in seconds.
 * (Options 4 and 5 of rounds 1-2 -- a persistent one-launch-per-series kernel and a one-launch-per-observation kernel, both
 * bit-identical and both measured slower than two launches per observation -- were removed in round 3; docs/history/LABNOTES_rounds1-3.md (sections 5b-5c) keeps
 * the record, git history the code.) */
#define CSSM_OPT_WHOLE_TILES 6
/* CSSM_OPT_GROUP_SUMS (default 1; a verification switch: results are bit-identical either way).  Where k_propagate runs one
 * fused-sums block per unit of a single-GPU cloud (every cloud of 2^20 particles or more that is not on geometry 3; smaller ones
 * under CSSM_OPT_WHOLE_TILES = 1 or 2 from 64 units on) its blocks also add their unit sums, limb by limb, to the sums of groups
 * of 32 units, and k_offspring's blocks read 32 group sums + the 32 unit sums of their own group instead of all 1024 unit sums
 * (16 KiB per block through the L2s).  0 = every block totals every unit sum, as before. */
#define CSSM_OPT_GROUP_SUMS 7
/* CSSM_OPT_SPECIALISE (default 1; results are bit-identical either way): the fused kernel holds the model's STRUCTURE -- per latent
 * component its SDE, its place in the f map, where its leaf ends -- and its observation model at compile time, as the reference's
 * models are composed at compile time (model/Model.scala:110-136): ahead-of-time instantiations for BASELINE's configurations, and for
 * every other model (d <= 12) one compiled at RUN TIME with hipRTC from the library's own kernel sources (~1 s per kernel the first time
 * a structure is seen; kept on disk afterwards: $CSSM_RTC_CACHE, else rtc_cache/ next to the library).  If the runtime compiler is
 * unavailable one line on stderr says so and the structure-as-data kernel runs; environment CSSM_RTC=0 does the same on purpose.
 * 0: always the structure-as-data kernel (bench.py reports its roofline fraction beside the specialised kernel's); 2 (verification):
 * the run-time-compiled kernel also where an ahead-of-time instantiation exists (tests compare the two). */
#define CSSM_OPT_SPECIALISE 8
/* CSSM_OPT_LOOP_EVENTS (default 0): the batch drivers record a HIP event before the first and behind the last per-observation kernel of a
 * call, which cssm_pf_last_loop_ms then reads.  Off by default: the two event packets cost a 20-observation call about 5 us (the one
 * between the last kernel and the call's closing kernel holds the queue for ~4 us). */
#define CSSM_OPT_LOOP_EVENTS 9
/* CSSM_OPT_WAVE_SUMS (default 1; a verification switch: results are bit-identical either way).  Single GPU, systematic resampling, clouds of
 * 2^20 particles and more on the tile-after-tile launch: every wave of the fused kernel owns a contiguous quarter of its block's unit and
 * stores the exact fixed-point sum of its weights; the resampling kernel (k_offspring_wave) takes a wave's prefix from those sums and the
 * group sums and runs the rest of its prefix arithmetic in fp64 -- no conversion to the 2^-96 grid and no 128-bit scan per weight, one block
 * barrier per block; the contract's exact predicate where the fp64 estimate is within its error band of a slot boundary, as before.
 * The mapping costs the fused kernel 1-4 % at d <= 2 and up to 10 % at d = 3 on the fastest boxes (a block streams four ranges per row
 * instead of one), the resampling kernel gains 14-16 % from 2^21 particles on and nothing at 2^20 (one tile per unit): 1 applies it to clouds
 * whose units have several tiles (more than 2^20 particles) of models with at most two latent components -- where the STEP gains --, 2
 * (verification, A/B) wherever the geometry allows, 0 = never: k_offspring_self (every weight converted and scanned in 128 bits). */
#define CSSM_OPT_WAVE_SUMS 10
int cssm_pf_set_option(cssm_pf* pf, int option, int value);
/* Counters of the run-time specialisation in this process: out4 = {kernels compiled, kernels loaded from the disk cache, launches of
 * run-time-compiled kernels, failures (each reported once on stderr)}. */
int cssm_rtc_info(uint64_t* out4);

/* Per-kernel device time, measured with HIP events recorded on the handle's stream directly
 * before and after every kernel launch of the batch loop (bench.py's `roofline` figure).
 * Enabling it inserts event records between kernels, so whole-loop throughput is measured with
 * profiling OFF.  The event packets themselves take time on the queue (about 2 us per bracketed launch on MI355X): the
 * caller can calibrate that in place as (loop time with profiling - loop time without) / bracketed launches, both loop
 * times from cssm_pf_last_loop_ms (bench.py does, and its figure then agrees with rocprofv3's kernel trace). */
#define CSSM_K_PROPAGATE 0   /* fused gather + propagate + weight + block max (+ fixed-point sums with CSSM_OPT_FUSED_SUMS) */
#define CSSM_K_TILE_SUMS 1   /* exp(w - level), fixed-point unit sums */
#define CSSM_K_OFFSPRING 2   /* unit prefix, ll / ess, cumulative weights -> end slots -> ancestor indices */
#define CSSM_K_REDUCE 3      /* k_reduce_units: the sums of k_propagate's single-tile blocks -> unit sums (large clouds) */
#define CSSM_K_PACK 4        /* sharded: k_boundary_pack (segment headers + boundary rows of the single-collective exchange) */
#define CSSM_K_EXPAND 5      /* sharded: k_offspring_expand_spec (offspring of the own particles + expansion of the received rows) */
#define CSSM_K_COLLECTIVE 6  /* sharded, collectives issued by the library: the RCCL kernel(s) of one observation's exchange */
#define CSSM_PROFILE_NKERNELS 7
int cssm_pf_profile(cssm_pf* pf, int enable);
int cssm_pf_profile_read(cssm_pf* pf, double* total_ms, uint64_t* launches);
/* ---- diagnostics of the numerics contract --------------------------------------------------- */

/* Evaluates one function of include/cssm_numerics.h ON THE DEVICE for n arguments (so that a caller holding the host
 * build of the same header can check that both builds agree bit for bit; the parity tests do).
 *   CSSM_FN_EXP / CSSM_FN_LOG / CSSM_FN_LOG_UNIT: out[i] = f(x[i])
 *   CSSM_FN_SINCOS2PI: out[2i] = sin, out[2i+1] = cos of 2 pi x[i]
 *   CSSM_FN_FIX: out holds 2 n 64-bit words (lo, hi) of cssm_fix_from_double(x[i]), written as raw bits
 *   CSSM_FN_PAIRED_NORMALS: x = {seed, first global particle id, step, tag, d} as doubles (n = 5 is ignored: n_out
 *                           particles are drawn); out[i * d + k] = normal k of particle first + i
 * n_out = number of doubles `out` can hold. */
#define CSSM_FN_EXP 0
#define CSSM_FN_LOG 1
#define CSSM_FN_LOG_UNIT 2
#define CSSM_FN_SINCOS2PI 3
#define CSSM_FN_FIX 4
#define CSSM_FN_PAIRED_NORMALS 5
#define CSSM_FN_SINCOS_U24 6   /* out[2i] = sin, out[2i+1] = cos of 2 pi k / 2^24, k = (uint32_t)x[i] < 2^24 (cssm_sincos_u24) */
#define CSSM_FN_SQRT_RADIUS 7  /* out[i] = cssm_sqrt_radius(x[i]), x[i] = +-0 or in [2^-39, 55.5] */
int cssm_contract_eval(int device, int fn, const double* x, size_t n, double* out, size_t n_out);

/* On-box streaming ceiling: GB/s (read + write) of a plain 16-bytes-per-lane device-to-device copy of `bytes` bytes,
 * HIP-event timed over `reps` launches.  bench.py reports k_propagate's rate against it beside the 8 TB/s specification
 * ("measure an on-box copy ceiling too and report both fractions"). */
int cssm_diag_copy_ceiling(int device, size_t bytes, int reps, double* gbps_out);

/* ---- inspection (parity tests, `PfState.particles` on demand) ------------------------------ */

uint64_t cssm_pf_num_particles(const cssm_pf* pf); /* local particle count */
int32_t cssm_pf_dim(const cssm_pf* pf);            /* total latent dimension d */

/* Current cloud, i.e. the RESAMPLED particles of the last step (PfState.particles,
 * model/ParticleFilter.scala:35), SoA: out[k*N + i] = component k of particle i.  Host ptr. */
int cssm_pf_get_particles(cssm_pf* pf, double* out_dN);
/* Ancestor indices of the last resampling (slot i <- particle anc[i]); host pointer, N. */
int cssm_pf_get_ancestors(cssm_pf* pf, uint32_t* out_N);
/* Log-weights of the last weighted step, before resampling (model/ParticleFilter.scala:123) -- where the handle keeps them: LGCP
 * filters, the multinomial resampler, cssm_pf_propagate, CSSM_OPT_FUSED_SUMS = 0, an observation that was redone relative to
 * the max.  Otherwise CSSM_ESTATE: the fused kernel stores the WEIGHTS in their place, see cssm_pf_get_weights. */
int cssm_pf_get_logw(cssm_pf* pf, double* out_N);
/* The weights w1_i = exp(min(w_i - c, 2^-20)) of the last weighted step -- what stepFilter hands its resampler as the second
 * argument, rescaled by the observation's reference level c instead of the max (model/ParticleFilter.scala:125-126;
 * include/cssm_numerics.h, "reference level") -- and *level_out = c (may be NULL).  CSSM_ESTATE where log-weights are kept. */
int cssm_pf_get_weights(cssm_pf* pf, double* out_N, double* level_out);
/* Propagated, not yet resampled particles of the last step (x1 at :118), SoA, host pointer. */
int cssm_pf_get_proposed(cssm_pf* pf, double* out_dN);

/* getIntervals (model/ParticleFilter.scala:415-424) of the current cloud, computed on the device so that
 * the streaming consumer (examples/Filtering.scala:29) never copies N particles back: per latent
 * component the mean (meanState, :465-479) and the two order statistics of getCredibleInterval
 * (:488-502: sorted(N - index - 1), sorted(index - 1), index = floor(interval*N)); for
 * eta = link(f(x, t)) the order statistics of getOrderStatistic (:455-460: sorted(N - index),
 * sorted(index)) and eta_of_mean = link(f(stateMean, t)) (:420).  Host pointers, d doubles each for the
 * state outputs; any may be NULL.  Order statistics are exact (radix selection); the means are plain
 * fp64 sums (agreement with the reference's sequential sum: ~1e-13 relative). */
int cssm_pf_summary(cssm_pf* pf, double interval, double* state_mean, double* state_lower, double* state_upper,
                    double* eta_of_mean, double* eta_lower, double* eta_upper);

/* FilterInterpolate (model/ParticleFilter.scala:273-311, ParticleFilter.interpolate :335-337): the filter whose
 * particles are whole paths, so that a weighted step resamples the paths and missing observations are
 * interpolated by the surviving lineages.  The forward pass keeps the history of clouds and ancestor arrays on the
 * device ((T+1) * N * (8 d + 4) bytes); the output is, for every time index s = 0..T (0 = initial cloud), the
 * summary examples/Interpolate.scala:42-44 forms from the transposed paths of the LAST state: mean, the
 * getCredibleInterval order statistics and eta (see cssm_pf_summary).  Arrays: (T+1)*d doubles (state_*),
 * T+1 doubles (eta_*); any may be NULL.  The handle must be re-initialised before further streaming calls.
 * flags: CSSM_INTERP_REFERENCE_PAIRING pairs output entry k with the cloud of time index T - k (evaluated with
 * the time of index k): what the example's `(interpolated, interpolated.last.particles.transpose).zipped` really
 * does, since the transposed paths run newest-first; 0 gives the chronological pairing the example intends. */
#define CSSM_INTERP_REFERENCE_PAIRING 1
int cssm_pf_interpolate(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T, double interval,
                        int flags, double* ll_out, double* state_mean, double* state_lower, double* state_upper,
                        double* eta_of_mean, double* eta_lower, double* eta_upper);

/* ---- stateless resampler: the `Resample[A]` seam (model/package.scala:23) ------------------ */

/* Resampling.systematicResampling (model/Resampling.scala:63-72) on host arrays: weights w[n]
 * (the unnormalised w1 = exp(w - max) the reference passes, :125-126), one uniform u in [0,1),
 * anc[n] receives the index of the particle that slot i copies.  Runs on `device`. */
int cssm_resample_systematic(const double* w, size_t n, double u, uint32_t* anc, int device);
/* The same seam for every resampler the library has: kind = CSSM_RESAMPLE_SYSTEMATIC (reads u; seed and step unused),
 * CSSM_RESAMPLE_STRATIFIED (model/Resampling.scala:78-86) or CSSM_RESAMPLE_MULTINOMIAL (:92-96), whose per-slot uniforms
 * come from the Philox streams of (seed, step) -- the ancestors a filter with that seed selects at observation `step`
 * for the same weights (u unused).  The reference draws them from unseeded global generators (:66, :83, :93). */
int cssm_resample(int kind, const double* w, size_t n, double u, uint64_t seed, uint32_t step, uint32_t* anc, int device);
/* EXTENSION -- not a drop-in for running reference code: Resampling.residualResampling (model/Resampling.scala:130-146) cannot run as
 * written (it hands Vector.range(1, m) with n weights to the multinomial resampler, indexes the particles with what comes back, and
 * exp-normalises weights that stepFilter has already exponentiated).  This is the resampler its scaladoc describes (:124-129): particle
 * i appears k_i = floor(n w_i / sum w) times, in particle order, and the remaining m = n - sum k_i slots are drawn by multinomial
 * resampling on the residual weights n w_i / sum w - k_i (the draws of cssm_resample(CSSM_RESAMPLE_MULTINOMIAL): one uniform per slot
 * from the Philox stream of (seed, step)).  anc[s] = the particle slot s takes: first the copies, then the m draws in draw order.
 * Arithmetic: csrc/cssm_residual.hip; oracle twin: oracle_resample_residual. */
int cssm_resample_residual(const double* w, size_t n, uint64_t seed, uint32_t step, uint32_t* anc, int device);

/* ---- sharded filter: stage calls between which the caller runs its collectives ------------ */
/*
 * EXACT exchange, one observation on R ranks -- LGCP series, forced, or the repetition of a series an
 * outlying observation voided:
 *   cssm_pf_shard_propagate     fused propagate + weight + fixed-point sums of exp(w - c) on the local shard, c being
 *                               the observation's reference level (cssm_numerics.h: known without any exchange);
 *                               5 x u64 out: S, S2 (2 words each), order key of the local max log-weight
 *   [all-gather of 5 x u64 per rank -- the only collective before the resampling exchange]
 *   cssm_pf_shard_offspring     global max -> was c usable?  If so: global cumulative weights -> end slot of every
 *                               local particle; ll, ess; for every destination rank q the range of local particles
 *                               that own at least one slot of q: first[R], count[R]; *redo_flag = 0.
 *                               If not (outlying observation; always for LGCP): *redo_flag = 1 and the caller runs
 *   cssm_pf_shard_sums          local sums again, relative to the max taken from the gathered words: 5 x u64
 *   [all-gather of 5 x u64 per rank]  and cssm_pf_shard_offspring once more
 *   [all-to-all of count[R], all-to-all-v of (d+1) doubles per particle]
 *   cssm_pf_shard_pack / _adopt pack the send ranges of the OTHER ranks (a rank's own range never
 *                               travels); adopt the rows received from lower (n_low) and higher
 *                               (n_high) ranks around the own range and expand to the N_local slots
 * All device work is enqueued on the stream given at creation; the only host reads are the
 * ones whose pointers are documented as host (the caller reads count[R] and the flag in one copy).
 */
int cssm_pf_shard_init(cssm_pf* pf, double t0);
int cssm_pf_shard_propagate(cssm_pf* pf, double t, double obs, int has_obs, uint64_t* sums5_dev);
int cssm_pf_shard_sums(cssm_pf* pf, const uint64_t* all_sums5_dev, int world, uint64_t* sums5_dev);
int cssm_pf_shard_offspring(cssm_pf* pf, const uint64_t* all_sums5_dev, int rank, int world,
                            int64_t* send_first_dev, int64_t* send_count_dev, uint64_t* redo_flag_dev);
int cssm_pf_shard_pack(cssm_pf* pf, int world, const int64_t* send_first_host,
                       const int64_t* send_count_host, int skip_rank, double* send_buf_dev);
int cssm_pf_shard_adopt(cssm_pf* pf, const double* recv_buf_dev, int64_t n_low, int64_t n_high,
                        int64_t self_first, int64_t self_count);
int cssm_pf_shard_result(cssm_pf* pf, double* ll_out, int32_t* ess_out);

/* A series known in advance: the records of all T observations are uploaded once (cssm_pf_shard_begin) and observation s is
 * propagated by cssm_pf_shard_propagate_at(s) (in order; sums5_dev as above, or NULL when the single-collective exchange
 * totals the sums itself).  cssm_pf_shard_status at the end: ll, ess and the sticky bits -- 4: the max ruled some
 * observation's reference level out (run the series again with the exact exchange), 8: a capacity miss that was not resumed;
 * need[s] = rows observation s needed (diagnostics). */
int cssm_pf_shard_begin(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T);
/* (measurement) What one block of every exchange launch of the peer-written exchange spent polling its peers since the cloud was drawn, as of
 * the last cssm_pf_shard_status: out4 = {exchanges counted, ticks the first offspring block waited for every rank's header words, ticks
 * the first expansion block waited for them, ticks it then waited for its neighbours' eager-rows flags}; ticks of the GPU's constant
 * 100 MHz clock.  At world 1 a block waits for its own launch's header block; across GPUs a slow link or a late peer shows up here and
 * not in the kernels' durations.  bench.py reports the per-exchange means (per_rank.header_wait_us, rows_wait_us). */
int cssm_pf_shard_wait_stats(cssm_pf* pf, uint64_t* out4);

/* T MORE observations of the sharded filter that is already running -- what cssm_pf_ll_filter_more is to a single-GPU handle
 * (Flow.scan(init)(stepFilter) handed the next T elements, model/ParticleFilter.scala:163-166): no new cloud, the first time
 * increment from the handle's clock, record s of the call = observation (observations so far + s) of the filter.  Steps,
 * exchanges, cssm_pf_shard_status and cssm_pf_shard_resume then address the call's records 0 .. T-1 as after _begin. */
int cssm_pf_shard_continue(cssm_pf* pf, const double* t, const double* y, const uint8_t* has_obs, size_t T);
int cssm_pf_shard_propagate_at(cssm_pf* pf, size_t s, uint64_t* sums5_dev);
int cssm_pf_shard_status(cssm_pf* pf, double* ll_out, int32_t* ess_out, uint32_t* bits_out, uint32_t* need, size_t T);
/* `filter` on shards (model/ParticleFilter.scala:152-158): with cssm_pf_shard_want_path(pf, 1) set BEFORE cssm_pf_shard_begin /
 * _continue, the rank that owns the GLOBAL slot sampleOne picks after an observation (Resampling.scala:151-154; the index is a
 * function of seed and observation, the same on every rank) records the state in that slot.  cssm_pf_shard_get_path copies
 * this rank's (T + 1) x d rows to the host: row s + 1 = the pick after observation s of the series (row 0: of the initial cloud;
 * all zero after _continue), rows of slots other ranks own are all-zero bits -- the caller combines the ranks' paths (exactly
 * one is non-zero per row; ShardedFilter.filter sums the 32-bit halves of the bit patterns). */
int cssm_pf_shard_want_path(cssm_pf* pf, int on);
int cssm_pf_shard_get_path(cssm_pf* pf, double* out_host, size_t T);

/* getIntervals (model/ParticleFilter.scala:415-424; cssm_pf_summary above) of the SHARDED cloud.  The order statistics are
 * ranks in the global cloud of n_global particles, so every radix-selection pass totals the ranks' byte histograms with one
 * all-reduce; the means are the all-reduced local sums over n_global.  Stage calls, the same on every rank:
 *   cssm_pf_shard_summary_begin(pf, interval, sums_dev, hist_dev)   keys of this rank's resampled cloud (d state rows +
 *                           eta = link(f(x, t))); sums_dev[d] <- local sums of the state components; hist_dev zeroed
 *   for shift = 56, 48, ..., 0:
 *     cssm_pf_shard_summary_hist(pf, shift, hist_dev)               local byte histograms of the still-matching keys
 *     [caller: all-reduce SUM of hist_dev, (d + 1) * 512 uint32]
 *     cssm_pf_shard_summary_pick(pf, shift, hist_dev)               every rank picks the same byte; hist_dev zeroed again
 *   [caller: all-reduce SUM of sums_dev]
 *   cssm_pf_shard_summary_finish(pf, sums_dev, ...)                 outputs as cssm_pf_summary (host pointers, may be NULL)
 * sums_dev (d doubles) and hist_dev ((d + 1) * 512 uint32) are DEVICE buffers of the caller (the ones its collective
 * works on); all launches go to the handle's stream.  Order statistics are exact (equal to the single-GPU filter's
 * over the concatenated shards); the means agree with it to ~1e-13 relative (another order of summation). */
int cssm_pf_shard_summary_begin(cssm_pf* pf, double interval, double* sums_dev, uint32_t* hist_dev);
int cssm_pf_shard_summary_hist(cssm_pf* pf, int shift, uint32_t* hist_dev);
int cssm_pf_shard_summary_pick(cssm_pf* pf, int shift, uint32_t* hist_dev);
int cssm_pf_shard_summary_finish(cssm_pf* pf, const double* sums_global_dev, double* state_mean, double* state_lower, double* state_upper,
                                 double* eta_of_mean, double* eta_lower, double* eta_upper);

/* SINGLE-COLLECTIVE exchange (every weighted observation of an ordinary series): ONE all-to-all per observation carries
 * the rank's 5 sum words (segment header, to every rank) AND its boundary particles -- to the rank below its first `cap`
 * particles, to the rank above its last `cap`, each with the inclusive prefix of its fixed-point weight within that block --
 * so no all-gather precedes it and the host reads nothing.  The receiver, holding every rank's sums after the exchange,
 * turns the prefixes into global end slots itself.  Slots of a rank owned neither by its own particles nor by the adjacent
 * ranks' boundary blocks raise sticky bit 8 (a capacity miss).
 *   cssm_pf_shard_spec_segment    doubles per pair of ranks for capacity `cap` (send / recv buffers: world segments)
 *   cssm_pf_shard_propagate_at    with sums5_dev = NULL
 *   [level from the global max instead (LGCP; repetition after an outlying observation): propagate_at with sums5_dev,
 *    all-gather of the 5 words, cssm_pf_shard_sums -- the sums are then relative to the level the gathered max selects]
 *   cssm_pf_shard_boundary_pack   a header for every destination, boundary rows for the two adjacent ranks
 *   (all-to-all of cssm_pf_shard_spec_segment doubles per pair)
 *   cssm_pf_shard_adopt_spec      offspring of the own particles, expansion of the received rows, coverage check
 * recv_buf_dev must stay untouched until the next propagate has run (ancestors point into it). */
int64_t cssm_pf_shard_spec_segment(const cssm_pf* pf, int64_t cap);
/* Particles per unit sum of this shard (a whole number of 1024-particle tiles).  A capacity that is a multiple of it lets
 * cssm_pf_shard_boundary_pack take the totals and prefixes of the boundary blocks from the unit sums instead of forming them
 * again (any capacity is correct; ShardedFilter rounds up to it). */
int64_t cssm_pf_shard_unit(const cssm_pf* pf);
int cssm_pf_shard_boundary_pack(cssm_pf* pf, int rank, int world, int64_t cap, double* send_buf_dev);
int cssm_pf_shard_adopt_spec(cssm_pf* pf, const double* recv_buf_dev, int rank, int world, int64_t cap);
/* A capacity miss is resumable.  The launch that found some rank's slots uncovered did nothing on any rank (every rank
 * reaches the same verdict from the segment headers), recorded the observation index and sticky bit 8, and every later
 * kernel of the series returned at once.  cssm_pf_shard_resume returns that index, clears the bit and rewinds the handle to
 * "that observation propagated, not yet resampled"; the host redoes its exchange with a larger capacity (boundary_pack,
 * all-to-all, adopt_spec) and continues the series behind it. */
int cssm_pf_shard_resume(cssm_pf* pf, uint32_t* fail_step_out);
/* ... and so is an observation whose reference level its max rules out (sticky bit 4: an outlying datum; the launch that found out
 * resampled nothing, recorded the observation, every later kernel returned at once).  cssm_pf_shard_resume_level returns its index,
 * clears the bit and rewinds the handle to "that observation NOT yet propagated" (its source cloud is untouched); the host runs it
 * again with the level taken from the all-gathered max (cssm_pf_shard_propagate_at with sums5_dev, cssm_pf_shard_sums, then the
 * exchange) and continues behind it -- also inside a continued series, which cannot be repeated from its start. */
int cssm_pf_shard_resume_level(cssm_pf* pf, uint32_t* fail_step_out);

/* The single-collective series with its collectives driven from INSIDE the library (RCCL over xGMI, resolved at run time
 * with dlopen: the copy already in the process, else librccl.so of the ROCm installation).  Per weighted observation s in
 * [s_begin, s_end): propagate_at -> boundary_pack -> ONE RCCL all-to-all -> adopt_spec, everything enqueued on the handle's
 * stream, no host wait.  single_collective = 1: equal-split ncclAllToAll of whole segments; 2: ncclAllToAllv in which only
 * the adjacent pairs exchange whole segments and every other pair the 12 header words (nothing else of those segments is
 * ever read; falls back to 1 when world <= 2 or the RCCL copy has no ncclAllToAllv); 3 = the same for any world, an error
 * without ncclAllToAllv (tests); + 4 (i.e. 5, 6, 7): the level of every observation comes from the GLOBAL max -- an
 * ncclAllGather of the ranks' 5 words and cssm_pf_shard_sums precede the all-to-all (LGCP series, whose level IS the max;
 * the repetition of a series an outlying observation voided): 2 collectives per observation, still no host read.
 * The communicator is made from an id that rank 0 creates and hands to the other ranks by
 * whatever channel the host has (bench.py: torch.distributed's object broadcast).  weighted[s] != 0: observation s
 * resamples.  Buffers (device): send / recv world * cssm_pf_shard_spec_segment doubles each; sums5 (5 words) / all_sums5
 * (5 * world words) are used by the + 4 modes only.
 * The library never waits without a bound for work that contains its collectives: cssm_pf_shard_status / _resume poll the
 * stream, watch the communicator's asynchronous error state and after CSSM_SHARD_TIMEOUT_S seconds (environment; default
 * 600) abort the communicator and return CSSM_ERCCL -- a rank that died or returned early surfaces as an error on every
 * rank instead of a hang.  A rank whose own RCCL call fails aborts the communicator before returning, for the same reason. */
typedef struct { char internal[128]; } cssm_rccl_id;   /* ncclUniqueId */
int cssm_rccl_available(void);
const char* cssm_rccl_library(void);   /* which librccl.so the library bound to (diagnostics) */
int cssm_rccl_unique_id(cssm_rccl_id* id_out);
int cssm_rccl_comm_create(const cssm_rccl_id* id, int world, int rank, int device, void** comm_out);
void cssm_rccl_comm_destroy(void* comm);
int cssm_pf_shard_series_rccl(cssm_pf* pf, void* comm, int rank, int world, size_t s_begin, size_t s_end,
                              const uint8_t* weighted, int64_t cap, uint64_t* sums5_dev, uint64_t* all_sums5_dev,
                              double* send_buf_dev, double* recv_buf_dev, int single_collective);

/* PEER-WRITTEN exchange: the same single-collective protocol without a collective.  Every rank owns two receive windows (they
 * alternate by exchange) and a flag per (window, source rank) in ONE slab of device memory that the other ranks map
 * (hipIpcOpenMemHandle across processes -- one process per GPU over xGMI peer access; the plain pointer where several shards share
 * a process, and for the rank itself).  k_boundary_pack writes segment (rank -> q) -- header, and for the two adjacent ranks the
 * boundary rows -- straight into q's window and, when the last of its blocks has made its stores visible at system scope, stores the
 * exchange number into q's flag (release); q's k_offspring_expand_spec polls the flags of all ranks (acquire, bounded: a rank that
 * never delivers raises a device error instead of hanging the GPU) before it reads a header.  Two launches per weighted
 * observation -- propagate, then pack + offspring + expansion in one (the pack blocks lead the grid) -- and nothing else: no RCCL
 * launch, no host wait.  The windows hold the
 * capacity they were set up for; an exchange that is resumed with a larger capacity, and observations whose level comes from the
 * global max, go through the collective exchange above (the two share every kernel and produce the same bits).
 *   cssm_pf_shard_peer_setup    allocate this rank's slab for (world, cap); *mine_out = what the other ranks need to map it
 *   [the host exchanges the handles: every rank receives all `world` of them, in rank order]
 *   cssm_pf_shard_peer_connect  map them, build the device table
 *   cssm_pf_shard_series_peer   observations [s_begin, s_end) of the resident series (cssm_pf_shard_begin / _continue), or stage
 *                               by stage: cssm_pf_shard_propagate_at(s, NULL), cssm_pf_shard_pack_peer, cssm_pf_shard_pack_rows_peer,
 *                               cssm_pf_shard_adopt_peer
 *   cssm_pf_shard_peer_close    unmap and free (also done by cssm_pf_destroy)
 * model/ParticleFilter.scala:116-132 is the step whose resampling this exchange completes across GPUs. */
typedef struct cssm_peer_handle {
  char ipc[64];          /* hipIpcMemHandle_t of the slab (valid if has_ipc) */
  uint64_t pid;          /* owner process: a rank of the same process uses local_ptr instead of the IPC handle */
  uint64_t local_ptr;    /* the slab's address in the owner's process */
  uint64_t bytes;        /* size of the slab: equal on all ranks (same world, capacity and latent dimension) */
  int32_t device, has_ipc;
} cssm_peer_handle;
int cssm_pf_shard_peer_setup(cssm_pf* pf, int rank, int world, int64_t cap, cssm_peer_handle* mine_out);
int cssm_pf_shard_peer_connect(cssm_pf* pf, const cssm_peer_handle* all_handles, int world);
/* One empty round of the protocol -- every rank writes a (non-zero) token into every rank's flags and waits, bounded, for everybody's
 * in its own -- called by all ranks at the same point right after connecting: a mapping, peer access or cross-GPU visibility that does
 * not work surfaces here, as CSSM_ESHARD on the host, before a series depends on it. */
int cssm_pf_shard_peer_handshake(cssm_pf* pf, uint32_t token);
/* (round 5) The handshake also carries a PAYLOAD: every rank stores 16 probe words (a function of token, source rank and index) at the
 * head of its segment in every peer's window 0 ahead of its token, the way the exchange's pack blocks store rows; a second launch behind
 * the handshake reads what the peers left in this rank's window with system-scope loads -- what every kernel of the library reads a
 * window with -- and CSSM_ESHARD names how many words read wrongly.  The same words read with PLAIN loads are counted too and reported
 * here (diagnostic: a line some cache of this GPU kept from an earlier round; not on any data path).  Callers run two rounds with
 * different tokens, a host barrier between them. */
uint32_t cssm_pf_shard_peer_probe_stale(const cssm_pf* pf);
void cssm_pf_shard_peer_close(cssm_pf* pf);
int cssm_pf_shard_pack_peer(cssm_pf* pf, int rank, int world, int64_t cap);
/* (round 5) EAGER ROWS + NEEDED ROWS.  A boundary block holds `cap` rows (sized for the worst observation, ~6 sqrt(N_global)), a typical
 * observation's neighbour needs a few hundred of them, and every row is a remote store over one xGMI link.  The rows next to the boundary
 * (CSSM_PEER_EAGER_ROWS, default 4096 = four tiles) are written AT ONCE, with a flag of their own that is long set when the reader gets to
 * them.  Rows beyond them travel only if the neighbour's slots need them: once the headers of all ranks are there the SENDER can tell
 * (end slot beyond its own last slot / run starting below its own first slot: the reader's arithmetic), so the pack's row blocks wait for
 * the headers, write those rows and publish their number behind a second flag -- which the reader waits for only when the eager rows do
 * not reach its first / last slot (it sees that from the eager rows).  In stage-by-stage use the rows beyond the eager ones are a stage of
 * their own: cssm_pf_shard_pack_peer (headers + eager rows) on every shard, cssm_pf_shard_pack_rows_peer on every shard,
 * cssm_pf_shard_adopt_peer on every shard.  CSSM_PEER_ALL_ROWS=1 (read when the handle is created), or an eager count >= the capacity:
 * every row travels at once, pack_rows_peer does nothing.
 * cssm_pf_shard_peer_rows: rows written for neighbours / neighbour segments / segments that needed rows beyond the eager ones, so far
 * (diagnostics; beyond_out may be NULL). */
int cssm_pf_shard_pack_rows_peer(cssm_pf* pf, int rank, int world, int64_t cap);
int cssm_pf_shard_peer_rows(cssm_pf* pf, uint64_t* rows_out, uint64_t* segments_out, uint64_t* beyond_out);
/* Diagnostics for a series that ended in "a rank's segment did not arrive": this rank's flag words of both windows (world x 96 uint32 each:
 * per source rank [0] header flag, [16] eager rows flag, [17] needed rows, [18] flag of the rows beyond the eager ones, [32..79] the 24
 * self-validating header words), its 256 local ticket words and the handle's exchange counter; returns the words written (< 0: error). */
int64_t cssm_pf_shard_peer_debug(cssm_pf* pf, uint32_t* out, size_t nwords);
int cssm_pf_shard_adopt_peer(cssm_pf* pf, int rank, int world, int64_t cap);
/* pack + adopt in ONE launch (the pack blocks lead the grid): for a rank that has its stream to itself, i.e. one process per GPU --
 * what cssm_pf_shard_series_peer enqueues.  Shards that share a stream use the stage calls, every stage on all shards before the next. */
int cssm_pf_shard_exchange_peer(cssm_pf* pf, int rank, int world, int64_t cap);
int cssm_pf_shard_series_peer(cssm_pf* pf, int rank, int world, size_t s_begin, size_t s_end, const uint8_t* weighted, int64_t cap);

/* ---- PMMH host loop ---------------------------------------------------------------------- */
/*
 * ParticleMetropolisHastings (model/PMMH.scala:68-81,114-123) with proposal
 * Parameters.perturb(delta) (model/Parameters.scala:65-67), a symmetric transition (0) and a
 * flat prior (0), as examples/DetermineParameters.scala:59,73 wires it.  theta0[n_theta] is
 * the flattened STORED parameter vector in Parameters.flattenParams order
 * (model/Parameters.scala:88-95: per leaf scale-if-present, then m0, c0, [mu|phi...] in
 * SdeParameter.flatten order).  Each iteration re-parameterises the handle, reseeds it with
 * cssm_pf_run_key(seed, iteration + 1), runs `filter`, and accepts iff log(u) < ll' - ll (initial ll = -1e99).
 * Outputs, one row per iteration (the stream drops the initial state, :97): ll[n_iters],
 * theta[n_iters * n_theta], accepted[n_iters] (running count), last_state[n_iters * d].
 */
int cssm_pmmh_run(cssm_pf* pf, const cssm_model_desc* desc, const double* theta0, size_t n_theta,
                  double delta, const double* t, const double* y, const uint8_t* has_obs, size_t T,
                  uint64_t seed, size_t n_iters, double* ll, double* theta, int32_t* accepted,
                  double* last_state);

/* ---- batched independent filters ------------------------------------------------------------ */
/*
 * B filters of ONE model structure -- the chains of a PMMH run (examples/DetermineParameters.scala:68-69 runs two under mapAsync(2)),
 * the parameter grid of a pilot run (model/Streaming.scala:38-39: mapAsyncUnordered(4)) -- advanced in lockstep: one launch per stage
 * for all of them (grid.y = the chain), enqueued by ONE host thread.  At N = 100 000 particles a single filter leaves most of the GPU
 * idle and a step is launch latency + one wave's dependent instruction stream; B chains cost little more than one.
 *   cssm_pfb_create          B handles of `desc`'s structure and n_particles each, on one stream
 *   cssm_pfb_filter          `filter` (model/ParticleFilter.scala:152-158) of chain k under descs[k] (same structure) and the Philox key
 *                            seeds[k]: ll_out[k], path_out[k][(T + 1) * d] (may be NULL), rc_out[k] = that chain's own status
 *                            (CSSM_ENONFINITE: its weights were unusable).  Per chain the bits of cssm_pf_reseed + cssm_pf_filter on a
 *                            handle of its own: a chain the batched launches do not serve (an observation ruled out of its reference
 *                            level, LGCP, another resampler) is run through that very driver.
 *   cssm_pmmh_run_batched    cssm_pmmh_run (model/PMMH.scala:68-81,114-123) for B chains in lockstep -- chain k from theta0[k * n_theta ..]
 *                            under seeds[k]: the same proposals, filter keys and decisions as B separate runs, every iteration's B
 *                            filters as one batch.  Outputs chain-major: ll[k * n_iters + it], theta[(k * n_iters + it) * n_theta + j],
 *                            accepted[k * n_iters + it], last_state[(k * n_iters + it) * d + j].
 *   cssm_pfb_chain           chain k as an ordinary handle (inspection: particles, ancestors, summaries); owned by the batch.
 */
typedef struct cssm_pfb cssm_pfb;
int cssm_pfb_create(const cssm_model_desc* desc, uint64_t n_particles, int n_chains, int device, cssm_pfb** out);
void cssm_pfb_destroy(cssm_pfb* b);
int cssm_pfb_num_chains(const cssm_pfb* b);
cssm_pf* cssm_pfb_chain(cssm_pfb* b, int k);
int cssm_pfb_filter(cssm_pfb* b, const cssm_model_desc* const* descs, const uint64_t* seeds, const double* t, const double* y,
                    const uint8_t* has_obs, size_t T, double* ll_out, double* path_out, int* rc_out);
int cssm_pmmh_run_batched(cssm_pfb* b, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, const double* t,
                          const double* y, const uint8_t* has_obs, size_t T, const uint64_t* seeds, size_t n_iters, double* ll, double* theta,
                          int32_t* accepted, double* last_state);
/* (round 5) ONE chain at two iterations per batch: a batch of three filters holds iteration i's proposal and BOTH candidates for iteration
 * i + 1 (its proposal from iteration i's proposal, and from the current parameters): the proposals and filter keys are functions of
 * (seed, iteration, parameters) alone, so they can be drawn and filtered before iteration i has decided.  Output identical to
 * cssm_pmmh_run(seed), bit for bit; `b` holds exactly three chains (a batch that fails is redone candidate by candidate in the sequential order: the same result on the error paths too).  At N = 100 000 a single filter leaves most of the GPU idle: three
 * cost little more than one (model/PMMH.scala:68-81). */
int cssm_pmmh_run_speculative(cssm_pfb* b, const cssm_model_desc* desc, const double* theta0, size_t n_theta, double delta, const double* t,
                              const double* y, const uint8_t* has_obs, size_t T, uint64_t seed, size_t n_iters, double* ll, double* theta,
                              int32_t* accepted, double* last_state);

/* Number of stored parameters of a descriptor and their copy-out / copy-in in flatten order. */
int cssm_desc_flatten(const cssm_model_desc* desc, double* theta, size_t cap, size_t* n_theta);

/* Diagnostic: the structure words the kernels branch on (one byte per latent component: SDE kind | place in the f map << 2 |
 * leaf ends << 4 | leftmost leaf << 5; four components per word, CSSM_MAX_DIM / 4 words) and the latent dimension.  The models
 * whose words are listed in csrc/cssm_prop.hip (KnownStructures) run kernels that hold them at compile time. */
int cssm_model_structure(const cssm_model_desc* desc, uint32_t* words_out, int32_t* d_out);

/* ---- errors / build info ------------------------------------------------------------------ */
const char* cssm_last_error(void);
const char* cssm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CSSM_PF_H */

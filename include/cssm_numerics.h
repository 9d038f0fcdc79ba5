/*
 * cssm_numerics.h -- the NUMERICS CONTRACT of libcssm_pf.
 *
 * Everything in this header is part of the drop-in boundary: a caller that wants to
 * reproduce the filter's random variates, weights and ancestor indices bit for bit (the
 * parity oracle under oracle/, a JVM-side replay, a different device) must evaluate exactly
 * these functions.  The header is plain C99 and is compiled unchanged by gcc (host) and by
 * hipcc for gfx950 (device); both builds MUST use -ffp-contract=off so that the only fused
 * multiply-adds are the explicit cssm_fma() calls below.
 *
 * Why the library does not simply call libm / the ROCm device libs: glibc and OCML round
 * exp/log/sin/cos differently in the last ulp, and one differing ulp in one weight can flip
 * one ancestor index, after which two filter runs only agree statistically.  SURVEY.md
 * section 7 ("hard parts") names this; the contract removes it.
 *
 * What the contract fixes:
 *   1. Philox4x32-7 (Salmon et al., SC'11; seven rounds since contract v5; since v6 one block yields TWO Box-Muller pairs),
 *      counter layout below.  It
 *      replaces the reference's unseeded global generators (breeze Rand at
 *      model/Sde.scala:13, scala.util.Random at model/Resampling.scala:66,152).
 *   2. cssm_exp / cssm_log / cssm_sincos2pi / Box-Muller: fixed polynomial evaluations
 *      (fdlibm-derived coefficients), < 2 ulp from the correctly rounded result.  Since contract v7 the Box-Muller angle -- 24 bits --
 *      goes through cssm_sincos_u24: a 256-entry (sin, cos) table rotated by the low 16 bits (absolute error <= 2^-52).
 *   3. Weight sums and the cumulative weight scan are accumulated in 128-bit FIXED POINT
 *      (96 fractional bits).  Integer addition is associative, so the sum and every prefix
 *      are independent of tile size, wave scan order, block order and GPU count, and the
 *      cumulative sums are monotone by construction.  The reference sums sequentially in
 *      fp64 (model/Resampling.scala:21-24,57); the contract value differs from that by at
 *      most N * 2^-53 relative (it is the more accurate of the two).
 */
#ifndef CSSM_NUMERICS_H
#define CSSM_NUMERICS_H

#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#elif !defined(CSSM_RTC_STDINT)   /* hipRTC builds have no system headers: the fixed-width types the sources use */
#define CSSM_RTC_STDINT 1
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
#ifndef offsetof
#define offsetof(type, member) __builtin_offsetof(type, member)
#endif
#endif

#if defined(__HIPCC__)
#define CSSM_HD __host__ __device__ inline
#else
#define CSSM_HD static inline
#endif

#ifdef __cplusplus
extern "C++" {
#endif

/* ------------------------------------------------------------------ bit casts, fma */

CSSM_HD uint64_t cssm_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
CSSM_HD double cssm_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
CSSM_HD double cssm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
CSSM_HD double cssm_sqrt(double x) { return __builtin_sqrt(x); } /* IEEE correctly rounded on both sides */
CSSM_HD double cssm_inf(void) { return cssm_u2d(0x7ff0000000000000ULL); }
CSSM_HD double cssm_nan(void) { return cssm_u2d(0x7ff8000000000000ULL); }
/* 2^k for k in [-1022, 1023] */
CSSM_HD double cssm_pow2i(int k) { return cssm_u2d((uint64_t)(k + 1023) << 52); }

/*
 * Primitives that have a one-instruction form on gfx950 and a portable form on the host.  Each pair returns the
 * SAME value for every argument the contract functions pass (stated per primitive), so the contract stays one
 * definition; tests/test_gpu_parity.py::test_contract_functions_on_the_device_match_the_host compares the two
 * builds function by function on dense and edge-case samples.
 */
#if defined(__HIP_DEVICE_COMPILE__)
#define CSSM_DEVICE_FORM 1
#else
#define CSSM_DEVICE_FORM 0
#endif
/* max(x, c) and min(x, c) against a finite non-zero constant c; a NaN x gives c (IEEE maxNum / minNum). */
CSSM_HD double cssm_max_c(double x, double c) {
#if CSSM_DEVICE_FORM
  return __builtin_fmax(x, c);   /* v_max_f64 */
#else
  return (x > c) ? x : c;
#endif
}
CSSM_HD double cssm_min_c(double x, double c) {
#if CSSM_DEVICE_FORM
  return __builtin_fmin(x, c);   /* v_min_f64 */
#else
  return (x < c) ? x : c;
#endif
}
/* p * 2^k for p in [0.5, 2) and k in [-1100, 1100], identical WHEREVER THE RESULT IS NORMAL OR OVERFLOWS (exact
 * scaling, +inf); the two forms may round a subnormal result differently, and cssm_exp discards those. */
CSSM_HD double cssm_scale2(double p, int k) {
#if CSSM_DEVICE_FORM
  return __builtin_ldexp(p, k);  /* v_ldexp_f64 */
#else
  const int k1 = k >> 1, k2 = k - k1;
  return (p * cssm_pow2i(k1)) * cssm_pow2i(k2);
#endif
}
/* fma(a, b, k) with a CONSTANT addend k (the Horner steps of the polynomials below).  The same IEEE operation on both
 * sides and at every level of the build knob CSSM_ASM_FMA_LEVEL (0: nowhere -- the default, 1: exp, 2: exp, log_unit,
 * sincos), which pins the device form to one v_fma_f64 with k in a scalar register pair.  Left to itself the compiler
 * (in the large kernels, not in small ones) keeps k in a vector register and emits v_mov_b64 + v_fmac_f64 per step.
 * Measured on MI355X, same box, us per observation (separate sums / fused sums): N = 2^20: level 0 40.2 / 39.0,
 * level 1 40.8 / 38.9, level 2 41.1 / 39.1; N = 2^24: 368 / 364, 357 / 356, 357 / 352.  Pinning frees 15-30 VGPRs and
 * gains 3 % at 2^24 but costs scalar-register pressure that loses 2 % at the size the benchmark is quoted on. */
#ifndef CSSM_ASM_FMA_LEVEL
#define CSSM_ASM_FMA_LEVEL 0
#endif
CSSM_HD double cssm_fma_k2(double a, double b, double k) {
#if CSSM_DEVICE_FORM && CSSM_ASM_FMA_LEVEL >= 2
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
#else
  return __builtin_fma(a, b, k);
#endif
}
CSSM_HD double cssm_fma_k(double a, double b, double k) {
#if CSSM_DEVICE_FORM && CSSM_ASM_FMA_LEVEL >= 1
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
#else
  return __builtin_fma(a, b, k);
#endif
}

/* fma(a, b, k) pinned, on the device, to ONE three-address v_fma_f64 with the coefficient k in a vector register: for some call
 * sites inside the fused kernel the compiler otherwise emits v_mov_b64 + v_fmac_f64 per Horner step (the two-address form needs
 * the addend in the destination, and the coefficient is still live).  The same IEEE operation everywhere.
 * (Measured and dropped: the coefficient in a SCALAR register pair -- 40 v_mov_b32 per thread of the fused kernel become s_mov_b32,
 * 64 -> 53 vector registers -- same-box A/B 16.10 vs 15.97 us at N = 2^20, 95.4 vs 94.0 us at d = 1, N = 2^24: no gain.) */
CSSM_HD double cssm_fma_kv(double a, double b, double k) {
#if CSSM_DEVICE_FORM
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(k));
  return d;
#else
  return __builtin_fma(a, b, k);
#endif
}

/* Round-to-nearest-even of t (|t| < 2^31) by the shifter trick: adding 1.5 * 2^52 leaves the integer in the low
 * mantissa bits (ulp = 1), in two's complement.  Plain IEEE additions: the same on every machine. */
#define CSSM_SHIFTER 0x1.8p52

/* ------------------------------------------------------------------ Philox4x32 (7 rounds; contract v5) */

typedef struct { uint32_t v[4]; } cssm_u32x4;

#define CSSM_PHILOX_M0 0xD2511F53u
#define CSSM_PHILOX_M1 0xCD9E8D57u
#define CSSM_PHILOX_W0 0x9E3779B9u
#define CSSM_PHILOX_W1 0xBB67AE85u

/* R rounds of Philox4x32 (Salmon et al., SC'11; the Random123 round function and Weyl key schedule). */
#if defined(__HIPCC__) && !defined(CSSM_PHILOX_NO_UNROLL)
#define CSSM_PHILOX_UNROLL _Pragma("unroll")
#elif defined(__HIPCC__)
#define CSSM_PHILOX_UNROLL _Pragma("unroll 1")
#else
#define CSSM_PHILOX_UNROLL
#endif
#define CSSM_PHILOX_BODY(R)                                                    \
  CSSM_PHILOX_UNROLL                                                           \
  for (int r = 0; r < (R); ++r) {                                              \
    uint64_t p0 = (uint64_t)CSSM_PHILOX_M0 * c.v[0];                           \
    uint64_t p1 = (uint64_t)CSSM_PHILOX_M1 * c.v[2];                           \
    cssm_u32x4 n;                                                              \
    n.v[0] = (uint32_t)(p1 >> 32) ^ c.v[1] ^ k0;                               \
    n.v[1] = (uint32_t)p1;                                                     \
    n.v[2] = (uint32_t)(p0 >> 32) ^ c.v[3] ^ k1;                               \
    n.v[3] = (uint32_t)p0;                                                     \
    c = n;                                                                     \
    k0 += CSSM_PHILOX_W0;                                                      \
    k1 += CSSM_PHILOX_W1;                                                      \
  }                                                                            \
  return c;
/* the 10-round generator of the Random123 known-answer vectors (tests: the round function and key schedule are right) */
CSSM_HD cssm_u32x4 cssm_philox4x32_10(cssm_u32x4 c, uint32_t k0, uint32_t k1) { CSSM_PHILOX_BODY(10) }
/* The generator of the contract (v5): SEVEN rounds.  Philox4x32-7 is the smallest round count the Random123 authors report as
 * passing BigCrush ("Crush-resistant"; 10 is their default for margin); every variate of the filter comes from it.  The
 * fused kernel is VALU-bound and Philox is its largest single item: three rounds fewer are 18 of ~60 VALU instructions
 * per block, 1.5 blocks per particle-step at d = 3.  (Contract v4 used 10 rounds; any fixed seed defines one realisation
 * of the same algorithm under either.) */
#define CSSM_PHILOX_ROUNDS 7
CSSM_HD cssm_u32x4 cssm_philox4x32(cssm_u32x4 c, uint32_t k0, uint32_t k1) { CSSM_PHILOX_BODY(CSSM_PHILOX_ROUNDS) }

/* Stream tags (bits 28..31 of counter word 3). */
#define CSSM_STREAM_STEP 0u   /* transition noise of filter step `step` (observation index) */
#define CSSM_STREAM_INIT 1u   /* initial-state draws (model/ParticleFilter.scala:105-108)   */
#define CSSM_STREAM_U 2u      /* the ONE uniform of systematic resampling (Resampling.scala:66) */
#define CSSM_STREAM_PICK 3u   /* sampleOne index of `filter` (Resampling.scala:151-154)     */
#define CSSM_STREAM_HOST 4u   /* host-side PMMH proposal / accept draws (PMMH.scala:70,74)  */
#define CSSM_STREAM_STRAT 5u  /* the N uniforms of stratified resampling (Resampling.scala:83), counter id = slot */
#define CSSM_STREAM_MULTI 6u  /* the N uniforms of multinomial resampling (Resampling.scala:93), counter id = slot */
#define CSSM_STREAM_KEY 7u    /* key derivation: the Philox key of filter run number `run` under user seed `seed` (cssm_derive_key) */

/*
 * Counter layout: word0/1 = stream id, word2 = step (observation index, 0-based; 0 for init),
 * word3 = tag<<28 | block.  Key = the 64-bit seed.  One Philox block yields TWO Box-Muller pairs (contract v6: pair P of a
 * stream is half P & 1 of its block P >> 1 -- words 0, 1 or words 2, 3; cssm_normal_pair64), i.e. four normals: normal q of
 * a stream is element q & 1 of pair q >> 1.  (Up to v5 a block fed one pair: 53 + 53 of its 128 bits.  Philox is a quarter
 * of the fused kernel's instructions; a pair now takes 40 + 24 bits, see cssm_normal_pair64.)
 * Stream ids are GLOBAL, so results do not depend on how particles are sharded over GPUs:
 *   - ordinary filter steps (CSSM_STREAM_STEP) and the initial draw (CSSM_STREAM_INIT): the two particles 2m and
 *     2m+1 share stream m = gid >> 1; of its normals, particle gid owns numbers q = (gid & 1) * d + k
 *     (k = latent component in Tree.flatten order) -- no variate is generated twice (a thread that owns both particles
 *     of a pair evaluates ceil(d / 2) blocks for their 2 d normals);
 *   - LGCP steps: stream id = gid, normal number q = substep * d + k;
 *   - resampling uniforms, sampleOne, host-side PMMH draws: see the stream tags above (PMMH proposal pair j of
 *     iteration `it`: half 0 of the block of (stream it, step j)).
 */
CSSM_HD uint64_t cssm_pair_stream(uint64_t gid) { return gid >> 1; }
CSSM_HD uint32_t cssm_pair_first(uint64_t gid, int d) { return (uint32_t)(gid & 1u) * (uint32_t)d; } /* q of component 0 */
CSSM_HD cssm_u32x4 cssm_philox_draw(uint64_t seed, uint64_t gid, uint32_t step, uint32_t tag, uint32_t block) {
  cssm_u32x4 c;
  c.v[0] = (uint32_t)gid;
  c.v[1] = (uint32_t)(gid >> 32);
  c.v[2] = step;
  c.v[3] = (tag << 28) | (block & 0x0FFFFFFFu);
  return cssm_philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* The Philox key of the `run`-th filter run a driver makes under one user seed (PMMH: one run per MCMC iteration,
 * model/PMMH.scala:71).  A PRF of (seed, run), NOT seed + run: chains started with adjacent seeds s and s + 1 would
 * otherwise replay each other's filter randomness shifted by one iteration (the reference's chains draw from
 * independent generator state, examples/DetermineParameters.scala:68-69). */
CSSM_HD uint64_t cssm_derive_key(uint64_t seed, uint64_t run) {
  const cssm_u32x4 b = cssm_philox_draw(seed, run, 0u, CSSM_STREAM_KEY, 0u);
  return (uint64_t)b.v[0] | ((uint64_t)b.v[1] << 32);
}

/* 53-bit integers from word pairs. */
CSSM_HD uint64_t cssm_bits53(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 21) | (lo >> 11); }
/* [0,1) with 53 bits, as scala.util.Random.nextDouble has: bits53 * 2^-53, formed from two exact
 * 32-bit conversions (hi * 2^-32 + (lo >> 11) * 2^-53: disjoint bit ranges, 53 bits, no rounding). */
CSSM_HD double cssm_u01(uint32_t hi, uint32_t lo) {
  return cssm_fma((double)hi, 0x1.0p-32, (double)(lo >> 11) * 0x1.0p-53);
}
/* (0,1] with 53 bits (safe argument of log): (bits53 + 1) * 2^-53, exact. */
CSSM_HD double cssm_u01_open0(uint32_t hi, uint32_t lo) { return cssm_u01(hi, lo) + 0x1.0p-53; }

/* ------------------------------------------------------------------ constant table */

/* CSSM_TAB: [0, 256) the log table of cssm_log_unit, 128 (invc, logc) pairs; [256, 768) the (sin, cos)(2 pi h / 256), h < 256, of
 * cssm_sincos_u24 (contract v7), each the correctly rounded double of the exact value (generated with rational arithmetic:
 * the angle reduced to the first octant exactly, Taylor series to 1e-70).  Kernels stage the table in LDS.  (All other constants
 * of the hot functions are literals: an LDS constant table measured slower on MI355X -- k_propagate 309 us vs 266 us
 * at N = 2^24 -- because table constants occupy VGPRs.) */
enum { CSSM_TAB_SIZE = 768, CSSM_TAB_SINCOS = 256 };
static const double CSSM_TAB[CSSM_TAB_SIZE] = {
  0x1.0000000000000p+0, 0x0.0p+0,
  0x1.fa11caa01fa12p-1, 0x1.7dc475f810a69p-7,
  0x1.f6310aca0dbb5p-1, 0x1.3cea44346a584p-6,
  0x1.f25f644230ab5p-1, 0x1.b9fc027af919ap-6,
  0x1.ee9c7f8458e02p-1, 0x1.1b0d98923d97fp-5,
  0x1.eae807aba01ebp-1, 0x1.58a5bafc8e4d3p-5,
  0x1.e741aa59750e4p-1, 0x1.95c830ec8e3f2p-5,
  0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0b56p-5,
  0x1.e01e01e01e01ep-1, 0x1.075983598e471p-4,
  0x1.dca01dca01dcap-1, 0x1.253f62f0a1417p-4,
  0x1.d92f2231e7f8ap-1, 0x1.42edcbea646eep-4,
  0x1.d5cac807572b2p-1, 0x1.60658a93750c4p-4,
  0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b12d0p-4,
  0x1.cf26e5c44bfc6p-1, 0x1.9ab42462033aep-4,
  0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb0eda0p-4,
  0x1.c8b265afb8a42p-1, 0x1.d4313d66cb35dp-4,
  0x1.c5894d10d4986p-1, 0x1.f0a30c01162a4p-4,
  0x1.c26b5392ea01cp-1, 0x1.0671512ca596fp-3,
  0x1.bf583ee868d8bp-1, 0x1.14785846742acp-3,
  0x1.bc4fd65883e7bp-1, 0x1.2266f190a5acdp-3,
  0x1.b951e2b18ff23p-1, 0x1.303d718e47fd5p-3,
  0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc62ap-3,
  0x1.b37484ad806cep-1, 0x1.4ba36f39a55e5p-3,
  0x1.b094b31d922a4p-1, 0x1.59338d9982085p-3,
  0x1.adbe87f94905ep-1, 0x1.66acd4272ad51p-3,
  0x1.aaf1d2f87ebfdp-1, 0x1.740f8f54037a3p-3,
  0x1.a82e65130e159p-1, 0x1.815c0a14357e9p-3,
  0x1.a574107688a4ap-1, 0x1.8e928de886d41p-3,
  0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7dfb85p-3,
  0x1.a01a01a01a01ap-1, 0x1.a8becfc882f19p-3,
  0x1.9d79f176b682dp-1, 0x1.b5b519e8fb5a6p-3,
  0x1.9ae24ea5510dap-1, 0x1.c2968558c18c2p-3,
  0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c5ddp-3,
  0x1.95cbb0be377aep-1, 0x1.dc1bca0abec7bp-3,
  0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa5a60p-3,
  0x1.90d4f120190d5p-1, 0x1.f550a564b7b37p-3,
  0x1.8e6527af1373fp-1, 0x1.00e6c45ad501dp-2,
  0x1.8bfce8062ff3ap-1, 0x1.071b85fcd590dp-2,
  0x1.899c0f601899cp-1, 0x1.0d46b579ab74bp-2,
  0x1.87427bcc092b9p-1, 0x1.136870293a8b0p-2,
  0x1.84f00c2780614p-1, 0x1.1980d2dd4236fp-2,
  0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a2f3p-2,
  0x1.8060180601806p-1, 0x1.2596010df763ap-2,
  0x1.7e225515a4f1dp-1, 0x1.2b9303ab89d25p-2,
  0x1.7beb3922e017cp-1, 0x1.31871c9544185p-2,
  0x1.79baa6bb6398bp-1, 0x1.3772662bfd85cp-2,
  0x1.77908119ac60dp-1, 0x1.3d54fa5c1f710p-2,
  0x1.756cac201756dp-1, 0x1.432ef2a04e813p-2,
  0x1.734f0c541fe8dp-1, 0x1.49006804009d0p-2,
  0x1.713786d9c7c09p-1, 0x1.4ec9732600269p-2,
  0x1.6f26016f26017p-1, 0x1.548a2c3add263p-2,
  0x1.6d1a62681c861p-1, 0x1.5a42ab0f4cfe2p-2,
  0x1.6b1490aa31a3dp-1, 0x1.5ff3070a793d4p-2,
  0x1.691473a88d0c0p-1, 0x1.659b57303e1f2p-2,
  0x1.6719f3601671ap-1, 0x1.6b3bb2235943dp-2,
  0x1.6524f853b4aa3p-1, 0x1.70d42e2789236p-2,
  0x1.63356b88ac0dep-1, 0x1.7664e1239dbcfp-2,
  0x1.614b36831ae94p-1, 0x1.7bede0a37afbfp-2,
  0x1.5f66434292dfcp-1, 0x1.816f41da0d495p-2,
  0x1.5d867c3ece2a5p-1, 0x1.86e919a330ba1p-2,
  0x1.5babcc647fa91p-1, 0x1.8c5b7c858b48bp-2,
  0x1.59d61f123ccaap-1, 0x1.91c67eb45a83ep-2,
  0x1.5805601580560p-1, 0x1.972a341135159p-2,
  0x1.56397ba7c52e2p-1, 0x1.9c86b02dc0862p-2,
  0x1.54725e6bb82fep+0, -0x1.23ec5991eba49p-2,
  0x1.52aff56a8054bp+0, -0x1.1e9e1678899f5p-2,
  0x1.50f22e111c4c5p+0, -0x1.1956d3b9bc2f9p-2,
  0x1.4f38f62dd4c9bp+0, -0x1.14167ef367784p-2,
  0x1.4d843bedc2c4cp+0, -0x1.0edd060b78082p-2,
  0x1.4bd3edda68fe1p+0, -0x1.09aa572e6c6d4p-2,
  0x1.4a27fad76014ap+0, -0x1.047e60cde83b7p-2,
  0x1.4880522014880p+0, -0x1.feb2233ea07cbp-3,
  0x1.46dce34596066p+0, -0x1.f474b134df228p-3,
  0x1.453d9e2c776cap+0, -0x1.ea4449f04aaf5p-3,
  0x1.43a2730abee4dp+0, -0x1.e020cc6235ab5p-3,
  0x1.420b5265e5951p+0, -0x1.d60a17f903514p-3,
  0x1.40782d10e6566p+0, -0x1.cc000c9db3c52p-3,
  0x1.3ee8f42a5af07p+0, -0x1.c2028ab17f9b5p-3,
  0x1.3d5d991aa75c6p+0, -0x1.b811730b823d4p-3,
  0x1.3bd60d9232955p+0, -0x1.ae2ca6f672bd8p-3,
  0x1.3a524387ac822p+0, -0x1.a454082e6ab03p-3,
  0x1.38d22d366088ep+0, -0x1.9a8778debaa3ap-3,
  0x1.3755bd1c945eep+0, -0x1.90c6db9fcbcdbp-3,
  0x1.35dce5f9f2af8p+0, -0x1.871213750e994p-3,
  0x1.34679ace01346p+0, -0x1.7d6903caf5acdp-3,
  0x1.32f5ced6a1dfap+0, -0x1.73cb9074fd14dp-3,
  0x1.3187758e9ebb6p+0, -0x1.6a399dabbd383p-3,
  0x1.301c82ac40260p+0, -0x1.60b3100b09474p-3,
  0x1.2eb4ea1fed14bp+0, -0x1.5737cc9018cddp-3,
  0x1.2d50a012d50a0p+0, -0x1.4dc7b897bc1c7p-3,
  0x1.2bef98e5a3711p+0, -0x1.4462b9dc9b3dcp-3,
  0x1.2a91c92f3c105p+0, -0x1.3b08b6757f2a7p-3,
  0x1.293725bb804a5p+0, -0x1.31b994d3a4f86p-3,
  0x1.27dfa38a1ce4dp+0, -0x1.28753bc11aba2p-3,
  0x1.268b37cd60127p+0, -0x1.1f3b925f25d44p-3,
  0x1.2539d7e9177b2p+0, -0x1.160c8024b27b0p-3,
  0x1.23eb79717605bp+0, -0x1.0ce7ecdccc28bp-3,
  0x1.22a0122a0122ap+0, -0x1.03cdc0a51ec0dp-3,
  0x1.21579804855e6p+0, -0x1.f57bc7d9005dbp-4,
  0x1.2012012012012p+0, -0x1.e3707ee30487bp-4,
  0x1.1ecf43c7fb84cp+0, -0x1.d179788219362p-4,
  0x1.1d8f5672e4abdp+0, -0x1.bf968769fca18p-4,
  0x1.1c522fc1ce059p+0, -0x1.adc77ee5aea8ep-4,
  0x1.1b17c67f2bae3p+0, -0x1.9c0c32d4d254dp-4,
  0x1.19e0119e0119ep+0, -0x1.8a6477a91dc29p-4,
  0x1.18ab083902bdbp+0, -0x1.78d02263d82d7p-4,
  0x1.1778a191bd684p+0, -0x1.674f089365a78p-4,
  0x1.1648d50fc3201p+0, -0x1.55e10050e0382p-4,
  0x1.151b9a3fdd5c9p+0, -0x1.4485e03dbdfb0p-4,
  0x1.13f0e8d344724p+0, -0x1.333d7f8183f4ap-4,
  0x1.12c8b89edc0acp+0, -0x1.2207b5c7854a1p-4,
  0x1.11a3019a74826p+0, -0x1.10e45b3cae829p-4,
  0x1.107fbbe011080p+0, -0x1.ffa6911ab9309p-5,
  0x1.0f5edfab325a2p+0, -0x1.dda8adc67ee59p-5,
  0x1.0e40655826011p+0, -0x1.bbcebfc68f424p-5,
  0x1.0d24456359e3ap+0, -0x1.9a187b573de81p-5,
  0x1.0c0a7868b4171p+0, -0x1.788595a3577c8p-5,
  0x1.0af2f722eecb5p+0, -0x1.5715c4c03cee1p-5,
  0x1.09ddba6af8360p+0, -0x1.35c8bfaa13069p-5,
  0x1.08cabb37565e2p+0, -0x1.149e3e4005a8dp-5,
  0x1.07b9f29b8eae2p+0, -0x1.e72bf2813ce6ap-6,
  0x1.06ab59c7912fbp+0, -0x1.a55f548c5c427p-6,
  0x1.059eea0727586p+0, -0x1.63d6178690bbep-6,
  0x1.04949cc1664c5p+0, -0x1.228fb1fea2e0ap-6,
  0x1.038c6b78247fcp+0, -0x1.c317384c75f0dp-7,
  0x1.02864fc7729e9p+0, -0x1.41929f968330cp-7,
  0x1.0182436517a37p+0, -0x1.8121214586b02p-8,
  0x1.0000000000000p+0, 0x0.0p+0,
  /* (sin, cos)(2 pi h / 256), h = 0 .. 255 */
  0x0.0p+0, 0x1.0000000000000p+0,
  0x1.92155f7a3667ep-6, 0x1.ffd886084cd0dp-1,
  0x1.91f65f10dd814p-5, 0x1.ff621e3796d7ep-1,
  0x1.2d52092ce19f6p-4, 0x1.fe9cdad01883ap-1,
  0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1,
  0x1.f564e56a9730ep-4, 0x1.fc26470e19fd3p-1,
  0x1.2c8106e8e613ap-3, 0x1.fa7557f08a517p-1,
  0x1.5e214448b3fc6p-3, 0x1.f8764fa714ba9p-1,
  0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1,
  0x1.c0b826a7e4f63p-3, 0x1.f38f3ac64e589p-1,
  0x1.f19f97b215f1bp-3, 0x1.f0a7efb9230d7p-1,
  0x1.111d262b1f677p-2, 0x1.ed740e7684963p-1,
  0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1,
  0x1.4135c94176601p-2, 0x1.e6288ec48e112p-1,
  0x1.58f9a75ab1fddp-2, 0x1.e212104f686e5p-1,
  0x1.7088530fa459fp-2, 0x1.ddb13b6ccc23cp-1,
  0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1,
  0x1.9ef7943a8ed8ap-2, 0x1.d4134d14dc93ap-1,
  0x1.b5d1009e15cc0p-2, 0x1.ced7af43cc773p-1,
  0x1.cc66e9931c45ep-2, 0x1.c954b213411f5p-1,
  0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1,
  0x1.f8ba4dbf89abap-2, 0x1.bd7c0ac6f952ap-1,
  0x1.073879922ffeep-1, 0x1.b728345196e3ep-1,
  0x1.11eb3541b4b23p-1, 0x1.b090a58150200p-1,
  0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1,
  0x1.26d054cdd12dfp-1, 0x1.a29a7a0462782p-1,
  0x1.30ff7fce17035p-1, 0x1.9b3e047f38741p-1,
  0x1.3affa292050b9p-1, 0x1.93a22499263fbp-1,
  0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1,
  0x1.4e6cabbe3e5e9p-1, 0x1.83b0e0bff976ep-1,
  0x1.57d69348ceca0p-1, 0x1.7b5df226aafafp-1,
  0x1.610b7551d2cdfp-1, 0x1.72d0837efff96p-1,
  0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1,
  0x1.72d0837efff96p-1, 0x1.610b7551d2cdfp-1,
  0x1.7b5df226aafafp-1, 0x1.57d69348ceca0p-1,
  0x1.83b0e0bff976ep-1, 0x1.4e6cabbe3e5e9p-1,
  0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1,
  0x1.93a22499263fbp-1, 0x1.3affa292050b9p-1,
  0x1.9b3e047f38741p-1, 0x1.30ff7fce17035p-1,
  0x1.a29a7a0462782p-1, 0x1.26d054cdd12dfp-1,
  0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1,
  0x1.b090a58150200p-1, 0x1.11eb3541b4b23p-1,
  0x1.b728345196e3ep-1, 0x1.073879922ffeep-1,
  0x1.bd7c0ac6f952ap-1, 0x1.f8ba4dbf89abap-2,
  0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2,
  0x1.c954b213411f5p-1, 0x1.cc66e9931c45ep-2,
  0x1.ced7af43cc773p-1, 0x1.b5d1009e15cc0p-2,
  0x1.d4134d14dc93ap-1, 0x1.9ef7943a8ed8ap-2,
  0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2,
  0x1.ddb13b6ccc23cp-1, 0x1.7088530fa459fp-2,
  0x1.e212104f686e5p-1, 0x1.58f9a75ab1fddp-2,
  0x1.e6288ec48e112p-1, 0x1.4135c94176601p-2,
  0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2,
  0x1.ed740e7684963p-1, 0x1.111d262b1f677p-2,
  0x1.f0a7efb9230d7p-1, 0x1.f19f97b215f1bp-3,
  0x1.f38f3ac64e589p-1, 0x1.c0b826a7e4f63p-3,
  0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3,
  0x1.f8764fa714ba9p-1, 0x1.5e214448b3fc6p-3,
  0x1.fa7557f08a517p-1, 0x1.2c8106e8e613ap-3,
  0x1.fc26470e19fd3p-1, 0x1.f564e56a9730ep-4,
  0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4,
  0x1.fe9cdad01883ap-1, 0x1.2d52092ce19f6p-4,
  0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5,
  0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6,
  0x1.0000000000000p+0, 0x0.0p+0,
  0x1.ffd886084cd0dp-1, -0x1.92155f7a3667ep-6,
  0x1.ff621e3796d7ep-1, -0x1.91f65f10dd814p-5,
  0x1.fe9cdad01883ap-1, -0x1.2d52092ce19f6p-4,
  0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4,
  0x1.fc26470e19fd3p-1, -0x1.f564e56a9730ep-4,
  0x1.fa7557f08a517p-1, -0x1.2c8106e8e613ap-3,
  0x1.f8764fa714ba9p-1, -0x1.5e214448b3fc6p-3,
  0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3,
  0x1.f38f3ac64e589p-1, -0x1.c0b826a7e4f63p-3,
  0x1.f0a7efb9230d7p-1, -0x1.f19f97b215f1bp-3,
  0x1.ed740e7684963p-1, -0x1.111d262b1f677p-2,
  0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2,
  0x1.e6288ec48e112p-1, -0x1.4135c94176601p-2,
  0x1.e212104f686e5p-1, -0x1.58f9a75ab1fddp-2,
  0x1.ddb13b6ccc23cp-1, -0x1.7088530fa459fp-2,
  0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2,
  0x1.d4134d14dc93ap-1, -0x1.9ef7943a8ed8ap-2,
  0x1.ced7af43cc773p-1, -0x1.b5d1009e15cc0p-2,
  0x1.c954b213411f5p-1, -0x1.cc66e9931c45ep-2,
  0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2,
  0x1.bd7c0ac6f952ap-1, -0x1.f8ba4dbf89abap-2,
  0x1.b728345196e3ep-1, -0x1.073879922ffeep-1,
  0x1.b090a58150200p-1, -0x1.11eb3541b4b23p-1,
  0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1,
  0x1.a29a7a0462782p-1, -0x1.26d054cdd12dfp-1,
  0x1.9b3e047f38741p-1, -0x1.30ff7fce17035p-1,
  0x1.93a22499263fbp-1, -0x1.3affa292050b9p-1,
  0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1,
  0x1.83b0e0bff976ep-1, -0x1.4e6cabbe3e5e9p-1,
  0x1.7b5df226aafafp-1, -0x1.57d69348ceca0p-1,
  0x1.72d0837efff96p-1, -0x1.610b7551d2cdfp-1,
  0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1,
  0x1.610b7551d2cdfp-1, -0x1.72d0837efff96p-1,
  0x1.57d69348ceca0p-1, -0x1.7b5df226aafafp-1,
  0x1.4e6cabbe3e5e9p-1, -0x1.83b0e0bff976ep-1,
  0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1,
  0x1.3affa292050b9p-1, -0x1.93a22499263fbp-1,
  0x1.30ff7fce17035p-1, -0x1.9b3e047f38741p-1,
  0x1.26d054cdd12dfp-1, -0x1.a29a7a0462782p-1,
  0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1,
  0x1.11eb3541b4b23p-1, -0x1.b090a58150200p-1,
  0x1.073879922ffeep-1, -0x1.b728345196e3ep-1,
  0x1.f8ba4dbf89abap-2, -0x1.bd7c0ac6f952ap-1,
  0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1,
  0x1.cc66e9931c45ep-2, -0x1.c954b213411f5p-1,
  0x1.b5d1009e15cc0p-2, -0x1.ced7af43cc773p-1,
  0x1.9ef7943a8ed8ap-2, -0x1.d4134d14dc93ap-1,
  0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1,
  0x1.7088530fa459fp-2, -0x1.ddb13b6ccc23cp-1,
  0x1.58f9a75ab1fddp-2, -0x1.e212104f686e5p-1,
  0x1.4135c94176601p-2, -0x1.e6288ec48e112p-1,
  0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1,
  0x1.111d262b1f677p-2, -0x1.ed740e7684963p-1,
  0x1.f19f97b215f1bp-3, -0x1.f0a7efb9230d7p-1,
  0x1.c0b826a7e4f63p-3, -0x1.f38f3ac64e589p-1,
  0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1,
  0x1.5e214448b3fc6p-3, -0x1.f8764fa714ba9p-1,
  0x1.2c8106e8e613ap-3, -0x1.fa7557f08a517p-1,
  0x1.f564e56a9730ep-4, -0x1.fc26470e19fd3p-1,
  0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1,
  0x1.2d52092ce19f6p-4, -0x1.fe9cdad01883ap-1,
  0x1.91f65f10dd814p-5, -0x1.ff621e3796d7ep-1,
  0x1.92155f7a3667ep-6, -0x1.ffd886084cd0dp-1,
  0x0.0p+0, -0x1.0000000000000p+0,
  -0x1.92155f7a3667ep-6, -0x1.ffd886084cd0dp-1,
  -0x1.91f65f10dd814p-5, -0x1.ff621e3796d7ep-1,
  -0x1.2d52092ce19f6p-4, -0x1.fe9cdad01883ap-1,
  -0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1,
  -0x1.f564e56a9730ep-4, -0x1.fc26470e19fd3p-1,
  -0x1.2c8106e8e613ap-3, -0x1.fa7557f08a517p-1,
  -0x1.5e214448b3fc6p-3, -0x1.f8764fa714ba9p-1,
  -0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1,
  -0x1.c0b826a7e4f63p-3, -0x1.f38f3ac64e589p-1,
  -0x1.f19f97b215f1bp-3, -0x1.f0a7efb9230d7p-1,
  -0x1.111d262b1f677p-2, -0x1.ed740e7684963p-1,
  -0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1,
  -0x1.4135c94176601p-2, -0x1.e6288ec48e112p-1,
  -0x1.58f9a75ab1fddp-2, -0x1.e212104f686e5p-1,
  -0x1.7088530fa459fp-2, -0x1.ddb13b6ccc23cp-1,
  -0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1,
  -0x1.9ef7943a8ed8ap-2, -0x1.d4134d14dc93ap-1,
  -0x1.b5d1009e15cc0p-2, -0x1.ced7af43cc773p-1,
  -0x1.cc66e9931c45ep-2, -0x1.c954b213411f5p-1,
  -0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1,
  -0x1.f8ba4dbf89abap-2, -0x1.bd7c0ac6f952ap-1,
  -0x1.073879922ffeep-1, -0x1.b728345196e3ep-1,
  -0x1.11eb3541b4b23p-1, -0x1.b090a58150200p-1,
  -0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1,
  -0x1.26d054cdd12dfp-1, -0x1.a29a7a0462782p-1,
  -0x1.30ff7fce17035p-1, -0x1.9b3e047f38741p-1,
  -0x1.3affa292050b9p-1, -0x1.93a22499263fbp-1,
  -0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1,
  -0x1.4e6cabbe3e5e9p-1, -0x1.83b0e0bff976ep-1,
  -0x1.57d69348ceca0p-1, -0x1.7b5df226aafafp-1,
  -0x1.610b7551d2cdfp-1, -0x1.72d0837efff96p-1,
  -0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1,
  -0x1.72d0837efff96p-1, -0x1.610b7551d2cdfp-1,
  -0x1.7b5df226aafafp-1, -0x1.57d69348ceca0p-1,
  -0x1.83b0e0bff976ep-1, -0x1.4e6cabbe3e5e9p-1,
  -0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1,
  -0x1.93a22499263fbp-1, -0x1.3affa292050b9p-1,
  -0x1.9b3e047f38741p-1, -0x1.30ff7fce17035p-1,
  -0x1.a29a7a0462782p-1, -0x1.26d054cdd12dfp-1,
  -0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1,
  -0x1.b090a58150200p-1, -0x1.11eb3541b4b23p-1,
  -0x1.b728345196e3ep-1, -0x1.073879922ffeep-1,
  -0x1.bd7c0ac6f952ap-1, -0x1.f8ba4dbf89abap-2,
  -0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2,
  -0x1.c954b213411f5p-1, -0x1.cc66e9931c45ep-2,
  -0x1.ced7af43cc773p-1, -0x1.b5d1009e15cc0p-2,
  -0x1.d4134d14dc93ap-1, -0x1.9ef7943a8ed8ap-2,
  -0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2,
  -0x1.ddb13b6ccc23cp-1, -0x1.7088530fa459fp-2,
  -0x1.e212104f686e5p-1, -0x1.58f9a75ab1fddp-2,
  -0x1.e6288ec48e112p-1, -0x1.4135c94176601p-2,
  -0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2,
  -0x1.ed740e7684963p-1, -0x1.111d262b1f677p-2,
  -0x1.f0a7efb9230d7p-1, -0x1.f19f97b215f1bp-3,
  -0x1.f38f3ac64e589p-1, -0x1.c0b826a7e4f63p-3,
  -0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3,
  -0x1.f8764fa714ba9p-1, -0x1.5e214448b3fc6p-3,
  -0x1.fa7557f08a517p-1, -0x1.2c8106e8e613ap-3,
  -0x1.fc26470e19fd3p-1, -0x1.f564e56a9730ep-4,
  -0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4,
  -0x1.fe9cdad01883ap-1, -0x1.2d52092ce19f6p-4,
  -0x1.ff621e3796d7ep-1, -0x1.91f65f10dd814p-5,
  -0x1.ffd886084cd0dp-1, -0x1.92155f7a3667ep-6,
  -0x1.0000000000000p+0, 0x0.0p+0,
  -0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6,
  -0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5,
  -0x1.fe9cdad01883ap-1, 0x1.2d52092ce19f6p-4,
  -0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4,
  -0x1.fc26470e19fd3p-1, 0x1.f564e56a9730ep-4,
  -0x1.fa7557f08a517p-1, 0x1.2c8106e8e613ap-3,
  -0x1.f8764fa714ba9p-1, 0x1.5e214448b3fc6p-3,
  -0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3,
  -0x1.f38f3ac64e589p-1, 0x1.c0b826a7e4f63p-3,
  -0x1.f0a7efb9230d7p-1, 0x1.f19f97b215f1bp-3,
  -0x1.ed740e7684963p-1, 0x1.111d262b1f677p-2,
  -0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2,
  -0x1.e6288ec48e112p-1, 0x1.4135c94176601p-2,
  -0x1.e212104f686e5p-1, 0x1.58f9a75ab1fddp-2,
  -0x1.ddb13b6ccc23cp-1, 0x1.7088530fa459fp-2,
  -0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2,
  -0x1.d4134d14dc93ap-1, 0x1.9ef7943a8ed8ap-2,
  -0x1.ced7af43cc773p-1, 0x1.b5d1009e15cc0p-2,
  -0x1.c954b213411f5p-1, 0x1.cc66e9931c45ep-2,
  -0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2,
  -0x1.bd7c0ac6f952ap-1, 0x1.f8ba4dbf89abap-2,
  -0x1.b728345196e3ep-1, 0x1.073879922ffeep-1,
  -0x1.b090a58150200p-1, 0x1.11eb3541b4b23p-1,
  -0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1,
  -0x1.a29a7a0462782p-1, 0x1.26d054cdd12dfp-1,
  -0x1.9b3e047f38741p-1, 0x1.30ff7fce17035p-1,
  -0x1.93a22499263fbp-1, 0x1.3affa292050b9p-1,
  -0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1,
  -0x1.83b0e0bff976ep-1, 0x1.4e6cabbe3e5e9p-1,
  -0x1.7b5df226aafafp-1, 0x1.57d69348ceca0p-1,
  -0x1.72d0837efff96p-1, 0x1.610b7551d2cdfp-1,
  -0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1,
  -0x1.610b7551d2cdfp-1, 0x1.72d0837efff96p-1,
  -0x1.57d69348ceca0p-1, 0x1.7b5df226aafafp-1,
  -0x1.4e6cabbe3e5e9p-1, 0x1.83b0e0bff976ep-1,
  -0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1,
  -0x1.3affa292050b9p-1, 0x1.93a22499263fbp-1,
  -0x1.30ff7fce17035p-1, 0x1.9b3e047f38741p-1,
  -0x1.26d054cdd12dfp-1, 0x1.a29a7a0462782p-1,
  -0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1,
  -0x1.11eb3541b4b23p-1, 0x1.b090a58150200p-1,
  -0x1.073879922ffeep-1, 0x1.b728345196e3ep-1,
  -0x1.f8ba4dbf89abap-2, 0x1.bd7c0ac6f952ap-1,
  -0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1,
  -0x1.cc66e9931c45ep-2, 0x1.c954b213411f5p-1,
  -0x1.b5d1009e15cc0p-2, 0x1.ced7af43cc773p-1,
  -0x1.9ef7943a8ed8ap-2, 0x1.d4134d14dc93ap-1,
  -0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1,
  -0x1.7088530fa459fp-2, 0x1.ddb13b6ccc23cp-1,
  -0x1.58f9a75ab1fddp-2, 0x1.e212104f686e5p-1,
  -0x1.4135c94176601p-2, 0x1.e6288ec48e112p-1,
  -0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1,
  -0x1.111d262b1f677p-2, 0x1.ed740e7684963p-1,
  -0x1.f19f97b215f1bp-3, 0x1.f0a7efb9230d7p-1,
  -0x1.c0b826a7e4f63p-3, 0x1.f38f3ac64e589p-1,
  -0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1,
  -0x1.5e214448b3fc6p-3, 0x1.f8764fa714ba9p-1,
  -0x1.2c8106e8e613ap-3, 0x1.fa7557f08a517p-1,
  -0x1.f564e56a9730ep-4, 0x1.fc26470e19fd3p-1,
  -0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1,
  -0x1.2d52092ce19f6p-4, 0x1.fe9cdad01883ap-1,
  -0x1.91f65f10dd814p-5, 0x1.ff621e3796d7ep-1,
  -0x1.92155f7a3667ep-6, 0x1.ffd886084cd0dp-1,
};
#define CSSM_LOG_TAB CSSM_TAB

/* ------------------------------------------------------------------ exp */

/*
 * exp(x): k = round(x/ln2), r = x - k*ln2 (two-constant Cody-Waite with fma), degree-13
 * Taylor polynomial on |r| <= ln2/2 (truncation 4e-18 relative), scaled by 2^k in two exact
 * steps.  Contract edge cases: x > 709.78 -> +inf; x < -708 -> +0 (results are never
 * subnormal, so the contract does not depend on denormal support); NaN -> NaN.
 */
CSSM_HD double cssm_exp(double x) {
  const double LOG2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
  const double LN2_LO = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
  /* straight-line evaluation on a clamped argument (a NaN is clamped to -745 here); edge cases selected at the end */
  const double xc = cssm_min_c(cssm_max_c(x, -745.0), 710.0);
  const double ts = cssm_fma(xc, LOG2E, CSSM_SHIFTER); /* k = the integer nearest xc * log2(e), ties to even */
  const double kd = ts - CSSM_SHIFTER;
  const int k = (int)(uint32_t)cssm_d2u(ts);
  double r = cssm_fma(-kd, LN2_HI, xc);
  r = cssm_fma(-kd, LN2_LO, r);
  double p = 1.0 / 6227020800.0; /* 1/13! */
  p = cssm_fma_kv(p, r, 1.0 / 479001600.0);
  p = cssm_fma_kv(p, r, 1.0 / 39916800.0);
  p = cssm_fma_kv(p, r, 1.0 / 3628800.0);
  p = cssm_fma_kv(p, r, 1.0 / 362880.0);
  p = cssm_fma_kv(p, r, 1.0 / 40320.0);
  p = cssm_fma_kv(p, r, 1.0 / 5040.0);
  p = cssm_fma_kv(p, r, 1.0 / 720.0);
  p = cssm_fma_kv(p, r, 1.0 / 120.0);
  p = cssm_fma_kv(p, r, 1.0 / 24.0);
  p = cssm_fma_kv(p, r, 1.0 / 6.0);
  p = cssm_fma(p, r, 0.5);
  p = cssm_fma(p, r, 1.0);
  p = cssm_fma(p, r, 1.0);
  /* x > 709.782712893384 (= 1024 ln 2): k = 1024 and r >= 0, so p >= 1 and the scaling overflows to +inf by itself in both
   * forms of cssm_scale2 -- no select needed (there was one until round 3; it never changed a value) */
  double res = cssm_scale2(p, k);
  res = (x < -708.0) ? 0.0 : res;
  res = (x != x) ? x : res;
  return res;
}

/* exp(a) for an argument KNOWN to be at most CSSM_REF_BELOW and not NaN (-inf allowed): the weight w1 = exp(w - level) of a
 * particle whose log-weight was already checked for NaN and clamped at the level.  The value of cssm_exp(a) for every such a;
 * what is left out are the selects that cannot fire (overflow, NaN, the upper clamp).  Device form: the Horner steps are
 * pinned to one three-address v_fma_f64 each with the coefficient in a vector register -- in the fused kernel the compiler
 * otherwise emits v_mov_b64 + v_fmac_f64 per step for THIS call (ten copies per weight; the exp of the Poisson density a few
 * lines earlier, same source, gets plain v_fma_f64). */
CSSM_HD double cssm_exp_le0(double a) {
  const double LOG2E = 1.44269504088896338700e+00;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double xc = cssm_max_c(a, -745.0);
  const double ts = cssm_fma(xc, LOG2E, CSSM_SHIFTER);
  const double kd = ts - CSSM_SHIFTER;
  const int k = (int)(uint32_t)cssm_d2u(ts);
  double r = cssm_fma(-kd, LN2_HI, xc);
  r = cssm_fma(-kd, LN2_LO, r);
  double p = 1.0 / 6227020800.0;
  p = cssm_fma_kv(p, r, 1.0 / 479001600.0);
  p = cssm_fma_kv(p, r, 1.0 / 39916800.0);
  p = cssm_fma_kv(p, r, 1.0 / 3628800.0);
  p = cssm_fma_kv(p, r, 1.0 / 362880.0);
  p = cssm_fma_kv(p, r, 1.0 / 40320.0);
  p = cssm_fma_kv(p, r, 1.0 / 5040.0);
  p = cssm_fma_kv(p, r, 1.0 / 720.0);
  p = cssm_fma_kv(p, r, 1.0 / 120.0);
  p = cssm_fma_kv(p, r, 1.0 / 24.0);
  p = cssm_fma_kv(p, r, 1.0 / 6.0);
  p = cssm_fma(p, r, 0.5);
  p = cssm_fma(p, r, 1.0);
  p = cssm_fma(p, r, 1.0);
  const double res = cssm_scale2(p, k);
  return (a < -708.0) ? 0.0 : res;
}

/* ------------------------------------------------------------------ log */

/*
 * log(x), the fdlibm e_log algorithm without its short-cut branches: x = 2^k * m,
 * m in [sqrt(1/2), sqrt(2)), f = m-1, s = f/(2+f), log(m) = f - (f^2/2 - s*(f^2/2 + R(s^2))).
 * One IEEE division (correctly rounded on x86 and in hipcc's default fp64 lowering).
 * x = 0 -> -inf, x < 0 -> NaN, +inf -> +inf, NaN -> NaN; subnormals are scaled by 2^54.
 */
CSSM_HD double cssm_log(double x) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01,
               LG3 = 2.857142874366239149e-01, LG4 = 2.222219843214978396e-01,
               LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
               LG7 = 1.479819860511658591e-01;
  uint64_t ux = cssm_d2u(x);
  int k = 0;
  if (x != x) return x;
  if (ux >> 63) return ((ux << 1) == 0) ? -cssm_inf() : cssm_nan();
  if (ux == 0) return -cssm_inf();
  if (ux == 0x7ff0000000000000ULL) return x;
  if ((ux >> 52) == 0) { /* subnormal */
    x *= 0x1.0p54;
    ux = cssm_d2u(x);
    k -= 54;
  }
  uint32_t hx = (uint32_t)(ux >> 32);
  k += (int)(hx >> 20) - 1023;
  hx &= 0x000fffffu;
  uint32_t i = (hx + 0x95f64u) & 0x100000u;
  ux = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ux & 0xffffffffULL);
  k += (int)(i >> 20);
  double m = cssm_u2d(ux);
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double dk = (double)k;
  double z = s * s;
  double w = z * z;
  double t1 = w * cssm_fma(w, cssm_fma(w, LG6, LG4), LG2);
  double t2 = z * cssm_fma(w, cssm_fma(w, cssm_fma(w, LG7, LG5), LG3), LG1);
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

/*
 * log(x) for x in [2^-53, 1] -- the Box-Muller argument -- without a division and without special
 * cases.  x = 2^e * f, f in [1,2); j = top 7 mantissa bits of f.  j < 64: m = f, k = e; j >= 64:
 * m = f/2 in [0.75, 1), k = e + 1.  Table entry j holds invc = RN(1/c_j) and logc = RN(-log(invc)) for
 * the centre c_j of m's interval (c = 1 exactly for the two intervals touching 1, so log is exact-ish
 * and never positive near x = 1).  r = m*invc - 1 (one fma), |r| <= 2^-7,
 * log(x) = (k ln2_hi + logc) + (r + r^2 P(r) + k ln2_lo), P = degree-6 Taylor tail of log1p.
 * The table (2 KiB) is passed by pointer: the host uses CSSM_LOG_TAB, kernels stage it in LDS.
 * Generated with 80-digit decimal arithmetic (tests/golden/make_golden.py documents the recipe).
 */


CSSM_HD double cssm_log_unit(double x, const double* tab) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const uint64_t ux = cssm_d2u(x);
  const int e = (int)(ux >> 52) - 1023;
  const uint32_t j = (uint32_t)(ux >> 45) & 127u;
  const int up = (int)(j >> 6);
  const double m = cssm_u2d((ux & 0x000fffffffffffffULL) | ((uint64_t)(0x3ff - up) << 52));
  const double kd = (double)(e + up);
  const double invc = tab[2 * j], logc = tab[2 * j + 1];
  const double r = cssm_fma(m, invc, -1.0);
  double p = -1.0 / 8.0;
  p = cssm_fma_k2(p, r, 1.0 / 7.0);
  p = cssm_fma_k2(p, r, -1.0 / 6.0);
  p = cssm_fma_k2(p, r, 1.0 / 5.0);
  p = cssm_fma(p, r, -1.0 / 4.0);
  p = cssm_fma_k2(p, r, 1.0 / 3.0);
  p = cssm_fma(p, r, -0.5);
  const double hi = cssm_fma(kd, LN2_HI, logc);
  const double lo = cssm_fma(r * r, p, kd * LN2_LO);
  return hi + (r + lo);
}

/* ------------------------------------------------------------------ sin/cos of 2*pi*u */

/*
 * (sin, cos)(2*pi*u) for u in [0,1).  4u is split EXACTLY into a quadrant q = round(4u) and
 * r in [-1/2,1/2]; x = RN(r*pi/2) is fed to the fdlibm kernel polynomials (< 2 ulp overall).
 * No large-argument reduction exists, so there is nothing to get wrong at large t: seasonal
 * phases (model/Model.scala:217-223) are reduced as frac(a*t/P) first, see cssm_seasonal_phase.
 */
CSSM_HD void cssm_sincos2pi(double u, double* sn, double* cs) {
  const double PIO2_HI = 1.57079632679489655800e+00; /* 0x3FF921FB54442D18 */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double t = 4.0 * u;
  int q = (int)(t + 0.5);
  double r = t - (double)q;
  const double x = r * PIO2_HI; /* the angle, rounded once: |error| < 1.2e-16 * |x| */
  const double z = x * x;
  /* sin kernel: x + x^3 (S1 + z (S2 + ...)) */
  const double rs = cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, S6, S5), S4), S3), S2), S1);
  const double s = cssm_fma(z * x, rs, x);
  /* cos kernel: 1 - z/2 + z^2 (C1 + z (C2 + ...)) */
  const double rc = cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double c = w + (((1.0 - w) - hz) + (z * z) * rc);
  /* quadrant rotation without branches: odd q swaps, bit 1 of q (of q+1) negates sin (cos) */
  const int swap = q & 1;
  const double ss = swap ? c : s;
  const double cc = swap ? s : c;
  *sn = cssm_u2d(cssm_d2u(ss) ^ ((uint64_t)(q & 2) << 62));
  *cs = cssm_u2d(cssm_d2u(cc) ^ ((uint64_t)((q + 1) & 2) << 62));
}

/*
 * Seasonal phase: the reference evaluates cos(2*pi/P * a * t) (model/Model.scala:218-221).
 * The contract evaluates cos(2*pi*frac(a*t/P)) -- the same angle with exact period
 * reduction -- through cssm_sincos2pi.
 */
CSSM_HD double cssm_seasonal_phase(double a, double t, double period) {
  double x = (a * t) / period;
  double fl = (double)(long long)x;
  if (fl > x) fl -= 1.0;
  return x - fl;
}

/*
 * sin and cos of the angle 2 pi k / 2^24, k < 2^24 (the 24 angle bits of a Box-Muller pair): the high 8 bits of k pick
 * (sin a, cos a), a = 2 pi h / 256, from the table; the low 16 bits are a rotation by delta = 2 pi l / 2^24 < 2 pi / 256 = 0.0246,
 * whose sine and cosine need three Taylor terms each (truncation < 4e-19 / 4e-18 relative); sin(a + delta) = sin a cos delta +
 * cos a sin delta.  Absolute error < 2^-52 (what a normal variate needs: the relative error of a value near a zero crossing is
 * larger -- the two products cancel there).  20 instructions against the 40 of the full-range cssm_sincos2pi, which stays for
 * arguments that are not 24-bit fractions (the seasonal phase).
 */
CSSM_HD void cssm_sincos_u24(uint32_t k, const double* tab, double* sn, double* cs) {
  const uint32_t h = (k >> 16) & 255u, l = k & 0xffffu;
  const double sa = tab[CSSM_TAB_SINCOS + 2u * h], ca = tab[CSSM_TAB_SINCOS + 2u * h + 1u];
  const double dl = (double)l * 0x1.921fb54442d18p-22; /* l * (2 pi * 2^-24), 2 pi = 0x1.921fb54442d18p+2 */
  const double z = dl * dl;
  const double sd = cssm_fma(dl * z, cssm_fma_k2(z, cssm_fma_k2(z, -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7), -0x1.5555555555555p-3), dl);
  const double cd = cssm_fma_k2(z, cssm_fma_k2(z, cssm_fma_k2(z, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5), -0.5), 1.0);
  *sn = cssm_fma(ca, sd, sa * cd);
  *cs = cssm_fma(-sa, sd, ca * cd);
}

/* ------------------------------------------------------------------ Box-Muller */

/* sqrt of the Box-Muller radius argument t = -2 log u1: +-0 or a number in [2^-39, 55.5] (u1 = m * 2^-40, 1 <= m <= 2^40).  The
 * correctly rounded value on both sides.  Device: the compiler's own expansion of an IEEE double-precision sqrt -- v_rsq_f64, two
 * coupled Newton steps on (g, h) = (t y, y / 2), two residual corrections -- WITHOUT what that expansion spends on arguments
 * this one never sees: the rescaling of inputs below 2^-767 (a compare, two selects, two ldexp) and the pass-through of
 * infinities and NaNs.  Over the stated range the two are the same instruction sequence, hence the same bits
 * (tests/test_gpu_parity.py::test_contract_functions_on_the_device_match_the_host compares them with the host's sqrt). */
CSSM_HD double cssm_sqrt_radius(double t) {
#if CSSM_DEVICE_FORM
  const double y = __builtin_amdgcn_rsq(t);
  double g = t * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, t);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, t);
  g = __builtin_fma(d, h, g);
  return (t == 0.0) ? t : g;
#else
  return __builtin_sqrt(t);
#endif
}

/* Two standard normals from HALF a Philox block (two 32-bit words a, b): r = sqrt(-2 log u1), (r cos, r sin)(2 pi u2).
 *   u1 = (a * 2^8 + (b & 255) + 1) * 2^-40 in (0, 1]: 40 bits for the radius -- the largest |z| is sqrt(80 ln 2) = 7.45
 *        (a 53-bit uniform reaches 8.57; the tail beyond 7.45 has probability 9e-14 per variate: one in ~400 series of
 *        2^24 particles x 500 observations x 3 components would have drawn a single such value);
 *   u2 = (b >> 8) * 2^-24 in [0, 1): 24 bits for the angle (1.7e7 directions).
 * Both conversions are exact.  `tab` = CSSM_LOG_TAB (host) or its copy in LDS (kernels). */
CSSM_HD void cssm_normal_pair64(uint32_t a, uint32_t b, const double* tab, double* z0, double* z1) {
  const double u1 = cssm_fma((double)a, 0x1.0p-32, (double)((b & 255u) + 1u) * 0x1.0p-40);
  const double t = -2.0 * cssm_log_unit(u1, tab); /* log_unit(x <= 1) <= 0 by construction (tests/test_numerics_contract.py): t >= 0 */
  const double r = cssm_sqrt_radius(t);
  double sn, cs;
  cssm_sincos_u24(b >> 8, tab, &sn, &cs);
  *z0 = r * cs;
  *z1 = r * sn;
}
/* pair `half` (0 / 1) of a block, and pair number `pair` of a stream (host code, the oracle; kernels evaluate a block once
 * for both of its pairs) */
CSSM_HD void cssm_normal_pair_half(cssm_u32x4 blk, int half, const double* tab, double* z0, double* z1) {
  cssm_normal_pair64(half ? blk.v[2] : blk.v[0], half ? blk.v[3] : blk.v[1], tab, z0, z1);
}
CSSM_HD void cssm_normal_pair_of(uint64_t seed, uint64_t stream, uint32_t step, uint32_t tag, uint32_t pair, const double* tab,
                                 double* z0, double* z1) {
  cssm_normal_pair_half(cssm_philox_draw(seed, stream, step, tag, pair >> 1), (int)(pair & 1u), tab, z0, z1);
}

/* ------------------------------------------------------------------ lgamma(k+1), integer k */

/*
 * log(k!) for the Poisson density (model/Model.scala:273, breeze Poisson.logProbabilityOf).
 * It depends only on the observation, so it is evaluated once per observation on the host.
 * k < 256: sequential sum of cssm_log(i); otherwise the Stirling series (error < 1e-17).
 */
CSSM_HD double cssm_lgamma_kp1(long long k) {
  if (k < 2) return 0.0;
  if (k < 256) {
    double s = 0.0;
    for (long long i = 2; i <= k; ++i) s += cssm_log((double)i);
    return s;
  }
  double n = (double)k + 1.0;
  double inv = 1.0 / n, inv2 = inv * inv;
  double series = inv * (1.0 / 12.0 - inv2 * (1.0 / 360.0 - inv2 * (1.0 / 1260.0 - inv2 * (1.0 / 1680.0))));
  return (n - 0.5) * cssm_log(n) - n + 0.91893853320467274178 + series;
}

/*
 * log Gamma(x) for real x > 0 (host-side, once per observation: the constants of the negative
 * binomial and Student-t densities, model/Model.scala:191,158).  Argument shifted to z >= 24 by the
 * recurrence, then the Stirling series; ~1e-15 relative.
 */
CSSM_HD double cssm_lgamma(double x) {
  if (!(x > 0.0)) return cssm_nan();
  double prod = 1.0;   /* x (x+1) ... (z-1) <= 24! : one logarithm instead of a sum of up to 24 */
  double z = x;
  while (z < 24.0) { prod *= z; z += 1.0; }
  const double shift = cssm_log(prod);
  const double inv = 1.0 / z, inv2 = inv * inv;
  const double series = inv * (1.0 / 12.0 - inv2 * (1.0 / 360.0 - inv2 * (1.0 / 1260.0 - inv2 * (1.0 / 1680.0 - inv2 * (1.0 / 1188.0)))));
  return ((z - 0.5) * cssm_log(z) - z + 0.91893853320467274178 + series) - shift;
}

/* ------------------------------------------------------------------ 128-bit fixed point */

typedef struct { uint64_t lo, hi; } cssm_u128;

#define CSSM_FIX_FRAC_BITS 96 /* 1.0 == {lo = 0, hi = 1<<32}; 32 integer bits => N < 2^32 */

CSSM_HD cssm_u128 cssm_u128_zero(void) { cssm_u128 r; r.lo = 0; r.hi = 0; return r; }
/* a + b mod 2^128.  Device form: the four-instruction carry chain.  Left to itself the compiler (ROCm 7.2) lowers the portable
 * form -- and unsigned __int128, and __builtin_addcll -- to two or three 64-bit adds plus a 64-bit compare and a select per
 * 128-bit add: 5-8 instructions at the 64-bit rate (~5 cycles per wave each on MI355X, tools/instr_rate.hip) instead of 4 at
 * the 32-bit carry rate (~2.4).  The fused kernel makes two such adds per particle, a wave scan six per lane. */
CSSM_HD cssm_u128 cssm_u128_add(cssm_u128 a, cssm_u128 b) {
#if CSSM_DEVICE_FORM
  uint32_t a0 = (uint32_t)a.lo, a1 = (uint32_t)(a.lo >> 32), a2 = (uint32_t)a.hi, a3 = (uint32_t)(a.hi >> 32);
  const uint32_t b0 = (uint32_t)b.lo, b1 = (uint32_t)(b.lo >> 32), b2 = (uint32_t)b.hi, b3 = (uint32_t)(b.hi >> 32);
  __asm__("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %5, vcc\n\tv_addc_co_u32 %2, vcc, %2, %6, vcc\n\t"
          "v_addc_co_u32 %3, vcc, %3, %7, vcc"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");
  cssm_u128 r;
  r.lo = (uint64_t)a0 | ((uint64_t)a1 << 32);
  r.hi = (uint64_t)a2 | ((uint64_t)a3 << 32);
  return r;
#else
  cssm_u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
#endif
}
CSSM_HD int cssm_u128_is_zero(cssm_u128 a) { return (a.lo | a.hi) == 0; }

/* floor(w * 2^96) for finite 0 <= w < 2^9 (weights are <= 1); negative, NaN, inf and w >= 2^9 map to 0.
 * Four 32-bit digits peeled off by truncation: every subtraction and every scaling by 2^32 is exact in fp64 (the
 * operand is a non-negative double below 2^32 and its integer part), so no rounding occurs anywhere. */
/* ... the digits themselves, for an argument KNOWN to be finite and in [0, 2^32) (the weights a kernel has just formed as
 * exp of a clamped non-positive number): the same value as cssm_fix_from_double without its range selects. */
CSSM_HD cssm_u128 cssm_fix_from_unit(double x) {
  cssm_u128 r;
#if CSSM_DEVICE_FORM
  /* v_fract_f64 = t - floor(t): for a non-negative t below 2^32 exactly the t - (double)(uint32_t)t of the portable form
   * (dropping integer bits never rounds), one instruction instead of two per digit */
  const uint32_t f3 = (uint32_t)x;
  double tf = __builtin_amdgcn_fract(x) * 0x1.0p32;
  const uint32_t f2 = (uint32_t)tf;
  tf = __builtin_amdgcn_fract(tf) * 0x1.0p32;
  const uint32_t f1 = (uint32_t)tf;
  tf = __builtin_amdgcn_fract(tf) * 0x1.0p32;
  const uint32_t f0 = (uint32_t)tf;
  r.hi = ((uint64_t)f3 << 32) | f2;
  r.lo = ((uint64_t)f1 << 32) | f0;
  return r;
#endif
  const uint32_t d3 = (uint32_t)x;                        /* integer part */
  double t = (x - (double)d3) * 0x1.0p32;
  const uint32_t d2 = (uint32_t)t;
  t = (t - (double)d2) * 0x1.0p32;
  const uint32_t d1 = (uint32_t)t;
  t = (t - (double)d1) * 0x1.0p32;
  const uint32_t d0 = (uint32_t)t;
  r.hi = ((uint64_t)d3 << 32) | d2;
  r.lo = ((uint64_t)d1 << 32) | d0;
  return r;
}
CSSM_HD cssm_u128 cssm_fix_from_double(double w) {
  cssm_u128 r;
  const int valid = (w >= 0.0) & (w < 512.0);            /* false for NaN */
  const double x = valid ? w : 0.0;
  const uint32_t d3 = (uint32_t)x;                        /* integer part */
  double t = (x - (double)d3) * 0x1.0p32;
  const uint32_t d2 = (uint32_t)t;
  t = (t - (double)d2) * 0x1.0p32;
  const uint32_t d1 = (uint32_t)t;
  t = (t - (double)d1) * 0x1.0p32;
  const uint32_t d0 = (uint32_t)t;
  r.hi = ((uint64_t)d3 << 32) | d2;
  r.lo = ((uint64_t)d1 << 32) | d0;
  return r;
}

CSSM_HD int cssm_clz64(uint64_t x) { return __builtin_clzll(x); }

/* Correctly rounded (nearest-even) conversion of the 128-bit INTEGER to double: normalise to 64
 * significant bits + sticky bit (64 > 53 + 2, so rounding the 64-bit value once is rounding the
 * exact value once), convert, scale by an exact power of two. */
CSSM_HD double cssm_u128_to_double(cssm_u128 a) {
  const int zero = (a.hi | a.lo) == 0;
  const int big = a.hi == 0;                       /* value fits the low word */
  const uint64_t h2 = big ? a.lo : a.hi;
  const uint64_t l2 = big ? 0 : a.lo;
  const int lz = cssm_clz64(h2 | (uint64_t)zero);  /* 0..63 (h2 != 0 unless the value is 0) */
  uint64_t top = (h2 << lz) | ((l2 >> 1) >> (63 - lz));
  const uint64_t rest = l2 << lz;
  top |= (rest != 0) ? 1u : 0u;
  const double d = (double)top;                    /* in [2^63, 2^64] */
  const double v = d * cssm_pow2i((big ? 0 : 64) - lz);
  return zero ? 0.0 : v;
}

/* Same value scaled back to weight units (exact scaling by 2^-96). */
CSSM_HD double cssm_fix_to_double(cssm_u128 a) { return cssm_u128_to_double(a) * 0x1.0p-96; }

/* ------------------------------------------------------------------ systematic grid */

/* k_i = (u + i) / n exactly as model/Resampling.scala:69 (i widened to double). */
CSSM_HD double cssm_sys_grid(double u, uint64_t i, double nd) { return (u + (double)i) / nd; }

/*
 * cnt(C) = #{ i in [0,n) : k_i <= C }: the number of resampling slots whose grid point is at
 * or below the cumulative weight C.  k_i is non-decreasing in i, so an fp estimate plus an
 * exact fix-up with the literal predicate gives the exact count.  Particle j owns the slots
 * [cnt(C_{j-1}), cnt(C_j)), which is `TreeMap.from(k).head` (model/Resampling.scala:41-42)
 * with first-key-wins on equal cumulative weights.
 */
CSSM_HD uint64_t cssm_sys_count(double C, double u, uint64_t n) {
  const double nd = (double)n;
  const uint32_t nn = (uint32_t)n;                 /* n < 2^32 */
  const double est = C * nd - u;
  uint32_t c = (est > -1.0) ? ((est >= nd) ? nn : (uint32_t)(est + 1.0)) : 0u;
  c = (c > nn) ? nn : c;
  while (c < nn && (u + (double)c) / nd <= C) ++c;
  while (c > 0 && (u + (double)(c - 1)) / nd > C) --c;
  return c;
}

/*
 * The same count when n is a power of two: (u + i) / n == (u + i) * (1/n) bit for bit (scaling by a
 * power of two is exact; u + i >= 0 and n < 2^32 keep the product normal or zero), so the
 * division is replaced by a multiplication.  inv_n = 1.0 / n.
 */
CSSM_HD uint64_t cssm_sys_count_pow2(double C, double u, uint64_t n, double inv_n) {
  const double nd = (double)n;
  const uint32_t nn = (uint32_t)n;
  const double est = C * nd - u;
  uint32_t c = (est > -1.0) ? ((est >= nd) ? nn : (uint32_t)(est + 1.0)) : 0u;
  c = (c > nn) ? nn : c;
  while (c < nn && (u + (double)c) * inv_n <= C) ++c;
  while (c > 0 && (u + (double)(c - 1)) * inv_n > C) --c;
  return c;
}

/* Stratified grid, model/Resampling.scala:82-83: k_i = (i + u_i) / n with one uniform per slot,
 * u_i = cssm_u01 of Philox(seed, id = i, step, CSSM_STREAM_STRAT).  k_i is strictly increasing in i. */
CSSM_HD double cssm_strat_grid(uint64_t seed, uint32_t step, uint64_t i, double nd) {
  const cssm_u32x4 b = cssm_philox_draw(seed, i, step, CSSM_STREAM_STRAT, 0);
  return ((double)i + cssm_u01(b.v[0], b.v[1])) / nd;
}
/* #{ i in [0,n) : k_i <= C } for the stratified grid (same role as cssm_sys_count). */
CSSM_HD uint64_t cssm_strat_count(double C, uint64_t seed, uint32_t step, uint64_t n) {
  const double nd = (double)n;
  const uint32_t nn = (uint32_t)n;
  const double est = C * nd;
  uint32_t c = (est > 0.0) ? ((est >= nd) ? nn : (uint32_t)est) : 0u;
  while (c < nn && cssm_strat_grid(seed, step, c, nd) <= C) ++c;
  while (c > 0 && cssm_strat_grid(seed, step, c - 1, nd) > C) --c;
  return c;
}
/* The uniform of multinomial draw i (breeze Multinomial.draw: the first index whose cumulative weight
 * reaches u * sum, model/Resampling.scala:92-96). */
CSSM_HD double cssm_multi_uniform(uint64_t seed, uint32_t step, uint64_t i) {
  const cssm_u32x4 b = cssm_philox_draw(seed, i, step, CSSM_STREAM_MULTI, 0);
  return cssm_u01(b.v[0], b.v[1]);
}

/* ------------------------------------------------------------------ order-preserving key */

/* Monotone map double -> uint64 so that the global max of log-weights is an integer atomicMax. */
CSSM_HD uint64_t cssm_order_key(double x) {
  uint64_t u = cssm_d2u(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
CSSM_HD double cssm_order_unkey(uint64_t k) {
  return cssm_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k);
}

/* ------------------------------------------------------------------ reference level of a weighted step */

/* The reference's stepFilter rescales the log-weights by their maximum, w1 = exp(w - max)
 * (model/ParticleFilter.scala:124-125), which on a GPU costs a grid-wide dependency (and on several GPUs a
 * collective) between the log-densities and their sums.  The contract instead rescales by a level c that is
 * known BEFORE any particle is weighted: an upper bound of the observation log-density over gamma (its
 * supremum where that is simple), a function of the observation and the observation parameters alone.  The max
 * is still found (an integer atomicMax on the side) and decides afterwards whether c was usable:
 *
 *     ref = c   if  -CSSM_REF_BELOW <= c - max <= CSSM_REF_ABOVE      (every w1 <= 1 + 2^-20, top weight keeps > 50 bits)
 *         = max otherwise (outlying observation, non-finite c): the sums are then formed again with max.
 *     (LGCP: c is predicted from the previous event's max, see cssm_ref_predict below.)
 *
 * ll += ref + log(mean(exp(w - ref))) is the same quantity for either level; only roundings differ.
 * kinds: the CSSM_OBS_* numbers of cssm_pf.h; p = the observation parameter as the density uses it (Gaussian
 * sd, Student-t v); df = Student-t degrees of freedom. */
/* c is an upper bound of every log-weight, so the max can exceed it by rounding only: BELOW is a rounding allowance, not a
 * range.  It bounds every w1 by exp(2^-20), hence S and S2 by N (1 + 2^-20) < 2^32 for every admissible N (<= 2^32 - 2^16):
 * the 32 integer bits of the fixed-point sums cannot overflow.  (It was 6.0, which admitted w1 up to 403 and so, in
 * principle, a silent wrap of S beyond N = 2^23.) */
#define CSSM_REF_BELOW 0x1.0p-20
#define CSSM_REF_ABOVE 32.0
CSSM_HD double cssm_ref_level(int kind, double y, double p, double df) {
  switch (kind) {
    case 0: case 3: case 4: {   /* Poisson (sup at lambda = k); NegBin and ZIP (k > 0) are mixtures/multiples of it */
      const long long k = (long long)y;
      if (k < 1) return 0.0;
      const double kd = (double)k;
      return (kd * cssm_log(kd) - kd) - cssm_lgamma_kp1(k);
    }
    case 1: return -cssm_log(2.5066282746310002 * p);                       /* Gaussian at gamma = y */
    case 5: return 0.0;                                                      /* Bernoulli: log-probabilities */
    case 6:                                                                  /* Student-t at gamma = y: (1/v) * -logNormalizer */
      return ((cssm_lgamma((df + 1.0) / 2.0) - cssm_lgamma(df / 2.0)) - 0.5 * cssm_log(3.14159265358979311600 * df)) / p;
    case 7: {                                                                /* Beta(a, 1): sup over a at a = -1/log y */
      if (!(y > 0.0 && y < 1.0)) return 0.0;
      const double L = cssm_log(y);
      return (-1.0 - L) + cssm_log(-1.0 / L);
    }
    default: return cssm_nan();                                              /* LGCP: gamma - hazard is unbounded */
  }
}
/* LGCP (contract v8).  gamma - hazard has no bound that depends on the observation alone, so up to v7 every LGCP event was
 * rescaled by its max -- a grid-wide dependency between the weights and their sums on one GPU (a pass of its own over the
 * log-weights), and on several GPUs an all-gather of the maxima ahead of the resampling exchange.  Since v8 the level of an
 * event is PREDICTED from the max of the weighted observation before it in the same filter:
 *
 *     c_s = cssm_ref_predict(max_{s-1}) = max_{s-1} + CSSM_REF_PREDICT_MARGIN,      ref_s = cssm_ref_choose(c_s, max_s)
 *
 * and NaN (always the max) where no weighted observation precedes -- the first one after the cloud was initialised or handed
 * back by a host resampler.  The max of an LGCP event moves by a few units between events (the bench workload: at most
 * +-2.5), so the prediction stands practically always and the weights' sums are formed inside the propagate kernel as for
 * every other observation model; a prediction the max rules out (cssm_ref_choose) is an outlying observation like any other:
 * its sums are formed again relative to the max.  The margin keeps the top weight at exp(-margin) ~ 2^-11.5 of the level: 84 of
 * the 96 fractional bits remain below it. */
#define CSSM_REF_PREDICT_MARGIN 8.0
CSSM_HD double cssm_ref_predict(double prev_max) { return prev_max + CSSM_REF_PREDICT_MARGIN; }   /* NaN in, NaN out */
/* The level actually used, given the maximum log-weight of the step. */
CSSM_HD double cssm_ref_choose(double c, double max) {
  const double gap = c - max;
  const int ok = (gap >= -CSSM_REF_BELOW) && (gap <= CSSM_REF_ABOVE);       /* false for NaN */
  return ok ? c : max;
}

#ifdef __cplusplus
}
#endif
#endif /* CSSM_NUMERICS_H */

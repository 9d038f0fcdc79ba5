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
 *   1. Philox4x32-10 (Salmon et al., SC'11), counter layout cssm_philox_ctr() below.  It
 *      replaces the reference's unseeded global generators (breeze Rand at
 *      model/Sde.scala:13, scala.util.Random at model/Resampling.scala:66,152).
 *   2. cssm_exp / cssm_log / cssm_sincos2pi / Box-Muller: fixed polynomial evaluations
 *      (fdlibm-derived coefficients), < 2 ulp from the correctly rounded result.
 *   3. Weight sums and the cumulative weight scan are accumulated in 128-bit FIXED POINT
 *      (96 fractional bits).  Integer addition is associative, so the sum and every prefix
 *      are independent of tile size, wave scan order, block order and GPU count, and the
 *      cumulative sums are monotone by construction.  The reference sums sequentially in
 *      fp64 (model/Resampling.scala:21-24,57); the contract value differs from that by at
 *      most N * 2^-53 relative (it is the more accurate of the two).
 */
#ifndef CSSM_NUMERICS_H
#define CSSM_NUMERICS_H

#include <stdint.h>

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

/* ------------------------------------------------------------------ Philox4x32-10 */

typedef struct { uint32_t v[4]; } cssm_u32x4;

#define CSSM_PHILOX_M0 0xD2511F53u
#define CSSM_PHILOX_M1 0xCD9E8D57u
#define CSSM_PHILOX_W0 0x9E3779B9u
#define CSSM_PHILOX_W1 0xBB67AE85u

CSSM_HD cssm_u32x4 cssm_philox4x32_10(cssm_u32x4 c, uint32_t k0, uint32_t k1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)CSSM_PHILOX_M0 * c.v[0];
    uint64_t p1 = (uint64_t)CSSM_PHILOX_M1 * c.v[2];
    cssm_u32x4 n;
    n.v[0] = (uint32_t)(p1 >> 32) ^ c.v[1] ^ k0;
    n.v[1] = (uint32_t)p1;
    n.v[2] = (uint32_t)(p0 >> 32) ^ c.v[3] ^ k1;
    n.v[3] = (uint32_t)p0;
    c = n;
    k0 += CSSM_PHILOX_W0;
    k1 += CSSM_PHILOX_W1;
  }
  return c;
}

/* Stream tags (bits 28..31 of counter word 3). */
#define CSSM_STREAM_STEP 0u   /* transition noise of filter step `step` (observation index) */
#define CSSM_STREAM_INIT 1u   /* initial-state draws (model/ParticleFilter.scala:105-108)   */
#define CSSM_STREAM_U 2u      /* the ONE uniform of systematic resampling (Resampling.scala:66) */
#define CSSM_STREAM_PICK 3u   /* sampleOne index of `filter` (Resampling.scala:151-154)     */
#define CSSM_STREAM_HOST 4u   /* host-side PMMH proposal / accept draws (PMMH.scala:70,74)  */

/*
 * Counter layout: word0/1 = GLOBAL particle id (so results do not depend on how particles
 * are sharded over GPUs), word2 = step (observation index, 0-based; 0 for init),
 * word3 = tag<<28 | substep<<8 | pair.  `pair` p serves latent components 2p and 2p+1;
 * `substep` is the LGCP sub-step (0 for ordinary steps).  Key = the 64-bit seed.
 */
CSSM_HD cssm_u32x4 cssm_philox_draw(uint64_t seed, uint64_t gid, uint32_t step, uint32_t tag,
                                    uint32_t substep, uint32_t pair) {
  cssm_u32x4 c;
  c.v[0] = (uint32_t)gid;
  c.v[1] = (uint32_t)(gid >> 32);
  c.v[2] = step;
  c.v[3] = (tag << 28) | ((substep & 0xFFFFFu) << 8) | (pair & 0xFFu);
  return cssm_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
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
  /* straight-line evaluation on a clamped argument; the edge cases are selected at the end */
  double xc = (x > 710.0) ? 710.0 : x;
  xc = (xc < -745.0) ? -745.0 : xc;
  xc = (x != x) ? 0.0 : xc;
  double t = xc * LOG2E;
  int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
  double kd = (double)k;
  double r = cssm_fma(-kd, LN2_HI, xc);
  r = cssm_fma(-kd, LN2_LO, r);
  double p = 1.0 / 6227020800.0; /* 1/13! */
  p = cssm_fma(p, r, 1.0 / 479001600.0);
  p = cssm_fma(p, r, 1.0 / 39916800.0);
  p = cssm_fma(p, r, 1.0 / 3628800.0);
  p = cssm_fma(p, r, 1.0 / 362880.0);
  p = cssm_fma(p, r, 1.0 / 40320.0);
  p = cssm_fma(p, r, 1.0 / 5040.0);
  p = cssm_fma(p, r, 1.0 / 720.0);
  p = cssm_fma(p, r, 1.0 / 120.0);
  p = cssm_fma(p, r, 1.0 / 24.0);
  p = cssm_fma(p, r, 1.0 / 6.0);
  p = cssm_fma(p, r, 0.5);
  p = cssm_fma(p, r, 1.0);
  p = cssm_fma(p, r, 1.0);
  int k1 = k >> 1;
  int k2 = k - k1;
  double res = (p * cssm_pow2i(k1)) * cssm_pow2i(k2);
  res = (x > 709.782712893384) ? cssm_inf() : res;
  res = (x < -708.0) ? 0.0 : res;
  res = (x != x) ? x : res;
  return res;
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
 * log(x) for x in [2^-53, 1] -- the Box-Muller argument -- without any special case: the same
 * algorithm and coefficients as cssm_log, straight-line (x is a positive normal number).
 */
CSSM_HD double cssm_log_unit(double x) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01,
               LG3 = 2.857142874366239149e-01, LG4 = 2.222219843214978396e-01,
               LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
               LG7 = 1.479819860511658591e-01;
  uint64_t ux = cssm_d2u(x);
  uint32_t hx = (uint32_t)(ux >> 32);
  int k = (int)(hx >> 20) - 1023;
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

/* ------------------------------------------------------------------ sin/cos of 2*pi*u */

/*
 * (sin, cos)(2*pi*u) for u in [0,1).  4u is split EXACTLY into a quadrant q = round(4u) and
 * r in [-1/2,1/2]; x = r*pi/2 is formed as a double-double and fed to the fdlibm kernels.
 * No large-argument reduction exists, so there is nothing to get wrong at large t: seasonal
 * phases (model/Model.scala:217-223) are reduced as frac(a*t/P) first, see cssm_seasonal_phase.
 */
CSSM_HD void cssm_sincos2pi(double u, double* sn, double* cs) {
  const double PIO2_HI = 1.57079632679489655800e+00; /* 0x3FF921FB54442D18 */
  const double PIO2_LO = 6.12323399573676603587e-17; /* 0x3C91A62633145C07 */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double t = 4.0 * u;
  int q = (int)(t + 0.5);
  double r = t - (double)q;
  double xh = r * PIO2_HI;
  double xl = cssm_fma(r, PIO2_HI, -xh) + r * PIO2_LO;
  double z = xh * xh;
  /* sin kernel */
  double v = z * xh;
  double rs = cssm_fma(z, cssm_fma(z, cssm_fma(z, cssm_fma(z, S6, S5), S4), S3), S2);
  double s = xh - ((z * (0.5 * xl - v * rs) - xl) - v * S1);
  /* cos kernel */
  double rc = z * cssm_fma(z, cssm_fma(z, cssm_fma(z, cssm_fma(z, cssm_fma(z, C6, C5), C4), C3), C2), C1);
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  double c = w + (((1.0 - w) - hz) + (z * rc - xh * xl));
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

/* ------------------------------------------------------------------ Box-Muller */

/* Two standard normals from one Philox block: r = sqrt(-2 log u1), (r cos, r sin)(2 pi u2). */
CSSM_HD void cssm_normal_pair(cssm_u32x4 b, double* z0, double* z1) {
  double u1 = cssm_u01_open0(b.v[0], b.v[1]);
  double u2 = cssm_u01(b.v[2], b.v[3]);
  double r = cssm_sqrt(-2.0 * cssm_log_unit(u1));
  double sn, cs;
  cssm_sincos2pi(u2, &sn, &cs);
  *z0 = r * cs;
  *z1 = r * sn;
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
CSSM_HD cssm_u128 cssm_u128_add(cssm_u128 a, cssm_u128 b) {
  cssm_u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
}
CSSM_HD int cssm_u128_is_zero(cssm_u128 a) { return (a.lo | a.hi) == 0; }

/* floor(w * 2^96) for finite 0 <= w < 2^9 (weights are <= 1); negative, NaN, inf and w >= 2^9 map
 * to 0.  Straight-line: the 53-bit significand m, seen as the 128-bit number m * 2^64, is shifted
 * right by k = 1043 - biased_exponent. */
CSSM_HD cssm_u128 cssm_fix_from_double(double w) {
  cssm_u128 r;
  const uint64_t u = cssm_d2u(w);
  const int e = (int)((u >> 52) & 0x7ff);
  const uint64_t m = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
  int k = 1043 - e;
  const int valid = ((u >> 63) == 0) & (e != 0) & (e != 0x7ff) & (k >= 11);
  k = (k > 127) ? 127 : k;
  k = (k < 11) ? 11 : k;
  const int big = k >= 64;
  const int sft = k & 63;
  const uint64_t down = m >> sft;
  const uint64_t up = (m << 1) << (63 - sft); /* m << (64 - sft) without a shift by 64 */
  r.hi = (valid && !big) ? down : 0;
  r.lo = valid ? (big ? down : up) : 0;
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

/* ------------------------------------------------------------------ order-preserving key */

/* Monotone map double -> uint64 so that the global max of log-weights is an integer atomicMax. */
CSSM_HD uint64_t cssm_order_key(double x) {
  uint64_t u = cssm_d2u(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
CSSM_HD double cssm_order_unkey(uint64_t k) {
  return cssm_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k);
}

#ifdef __cplusplus
}
#endif
#endif /* CSSM_NUMERICS_H */

// exp_bench.hip -- (round 6) what a table-driven exponential would buy, MEASURED: the contract's cssm_exp_le0 (ln 2 reduction, degree-13
// Taylor polynomial, two-step scaling) against a candidate "v9" form -- 2^(j/64) from a 64-entry (hi, lo) table in LDS, degree-5 or degree-6
// polynomial on |r| <= ln2/128, the scale added into the exponent field -- as isolated kernels: M exponentials per thread of independent
// arguments in (-40, 0], results summed.  Prints ns per exponential and wave, the ratio, and the candidate's error against glibc's exp in
// ulps.  Not part of the library: the candidate changes the bits of every weight (a numerics contract v9).
//   hipcc -O3 -std=c++17 -ffp-contract=off -mfma --offload-arch=gfx950 -I include -I composablestatespacemodels_amd/csrc tools/exp_bench.hip -o tools/exp_bench.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "cssm_device.hip.h"

#define NTAB 64
template <int DEG>
__device__ __forceinline__ double exp_tab(double a, const double* __restrict__ tab /* LDS: (lo, hi) pairs */) {
  const double INV = 0x1.71547652b82fep+6;        /* 64 / ln 2 */
  const double L_HI = 0x1.62e42fefa0000p-7;       /* ln 2 / 64, 33 significant bits */
  const double L_LO = 0x1.cf79abc9e3b3ap-46;
  const double xc = cssm_max_c(a, -745.0);
  const double ts = cssm_fma(xc, INV, CSSM_SHIFTER);
  const double kd = ts - CSSM_SHIFTER;
  const uint32_t k = (uint32_t)cssm_d2u(ts);
  double r = cssm_fma(-kd, L_HI, xc);
  r = cssm_fma(-kd, L_LO, r);
  const double2 t = *reinterpret_cast<const double2*>(tab + 2u * (k & (NTAB - 1u)));   // (tail, 2^(j/64))
  // scale = 2^(j/64) * 2^(k >> 6): the quotient added into the exponent field (results are never subnormal: the caller's select below)
  const int e = (int)k >> 6;
  const uint64_t sb = cssm_d2u(t.y) + ((uint64_t)(uint32_t)e << 52);
  const double scale = cssm_u2d(sb);
  const double r2 = r * r;
  double q;
  if (DEG == 5) {
    q = cssm_fma(cssm_fma(r, 1.0 / 120.0, 1.0 / 24.0), r2, cssm_fma(r, 1.0 / 6.0, 0.5));
  } else {
    q = cssm_fma(cssm_fma(cssm_fma(r, 1.0 / 720.0, 1.0 / 120.0), r, 1.0 / 24.0), r2, cssm_fma(r, 1.0 / 6.0, 0.5));
  }
  const double p = cssm_fma(q, r2, r + t.x);
  double res = cssm_fma(scale, p, scale);
  res = (a < -708.0) ? 0.0 : res;
  return res;
}

template <int MODE, int M>
__global__ __launch_bounds__(256) void k_bench(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ tabg, int reps) {
  __shared__ __attribute__((aligned(16))) double tab[2 * NTAB];
  if (threadIdx.x < 2 * NTAB) tab[threadIdx.x] = tabg[threadIdx.x];
  __syncthreads();
  double x[M];
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = in[g * M + i];
  double acc = 0.0;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const double a = x[i] - 1e-3 * (double)rep;
      acc += (MODE == 0) ? cssm_exp_le0(a) : (MODE == 5 ? exp_tab<5>(a, tab) : exp_tab<6>(a, tab));
    }
  }
  out[g] = acc;
}
template <int MODE>
__global__ void k_values(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ tabg, size_t n) {
  __shared__ __attribute__((aligned(16))) double tab[2 * NTAB];
  if (threadIdx.x < 2 * NTAB) tab[threadIdx.x] = tabg[threadIdx.x];
  __syncthreads();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = (MODE == 0) ? cssm_exp_le0(in[g]) : (MODE == 5 ? exp_tab<5>(in[g], tab) : exp_tab<6>(in[g], tab));
}

static double ulps(double got, double want) {
  if (want == 0.0) return got == 0.0 ? 0.0 : 1e9;
  int e; frexp(want, &e);
  return fabs(got - want) / ldexp(1.0, e - 53);
}

int main() {
  constexpr int M = 8;
  const int blocks = 256 * 20, reps = 200;
  const size_t nthreads = (size_t)blocks * 256, n = nthreads * M;
  std::vector<double> h(n), tabh(2 * NTAB);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = -40.0 * (double)(s >> 11) * 0x1.0p-53; }
  for (int j = 0; j < NTAB; ++j) {
    const long double T = exp2l((long double)j / NTAB);
    const double hi = (double)T;
    tabh[2 * j] = (double)(T - (long double)hi) / hi;     // the tail RELATIVE to hi: scale * (p + tail) = hi 2^e (1 + p) + lo 2^e to first order
    tabh[2 * j + 1] = hi;
  }
  double *din, *dout, *dtab, *dv;
  hipMalloc(&din, n * 8); hipMalloc(&dout, nthreads * 8); hipMalloc(&dtab, 2 * NTAB * 8); hipMalloc(&dv, n * 8);
  hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dtab, tabh.data(), 2 * NTAB * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double t_ms[3] = {0, 0, 0};
  for (int round = 0; round < 4; ++round) {
    for (int mode = 0; mode < 3; ++mode) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL((k_bench<0, M>), dim3(blocks), dim3(256), 0, 0, din, dout, dtab, reps);
      if (mode == 1) hipLaunchKernelGGL((k_bench<5, M>), dim3(blocks), dim3(256), 0, 0, din, dout, dtab, reps);
      if (mode == 2) hipLaunchKernelGGL((k_bench<6, M>), dim3(blocks), dim3(256), 0, 0, din, dout, dtab, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (round > 0) t_ms[mode] += ms / 3.0;
    }
  }
  const double per = 1e6 / ((double)n * reps);      // ms -> ns per exponential (all lanes of the GPU busy)
  printf("exponentials per launch %.3g; per exponential and GPU: contract cssm_exp_le0 %.4f ps, table degree 5 %.4f ps (x %.3f), table degree 6 %.4f ps (x %.3f)\n",
         (double)n * reps, t_ms[0] * per * 1e3, t_ms[1] * per * 1e3, t_ms[1] / t_ms[0], t_ms[2] * per * 1e3, t_ms[2] / t_ms[0]);
  const char* names[3] = {"contract cssm_exp_le0", "table, degree 5", "table, degree 6"};
  std::vector<double> v(n);
  for (int mode = 0; mode < 3; ++mode) {
    const dim3 g((unsigned)((n + 255) / 256));
    if (mode == 0) hipLaunchKernelGGL((k_values<0>), g, dim3(256), 0, 0, din, dv, dtab, n);
    if (mode == 1) hipLaunchKernelGGL((k_values<5>), g, dim3(256), 0, 0, din, dv, dtab, n);
    if (mode == 2) hipLaunchKernelGGL((k_values<6>), g, dim3(256), 0, 0, din, dv, dtab, n);
    hipMemcpy(v.data(), dv, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, sum = 0; size_t off = 0;
    for (size_t i = 0; i < n; i += 7) { const double u = ulps(v[i], exp(h[i])); worst = u > worst ? u : worst; sum += u; off += (u > 0.5); }
    printf("%-24s max error vs glibc exp %.3f ulp, mean %.4f, differing from the correctly rounded value in %.2f %% of %zu arguments in (-40, 0]\n",
           names[mode], worst, sum / (double)((n + 6) / 7), 100.0 * (double)off / (double)((n + 6) / 7), (n + 6) / 7);
  }
  return 0;
}

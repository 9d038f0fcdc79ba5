// tools/microbench.hip -- where do the cycles of k_propagate go?  Standalone (hipcc, no python):
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/microbench.hip -o /tmp/mb && /tmp/mb
// Times, at N particles, kernels that each do one more piece of the per-particle work and write one
// double per particle (so nothing is dead code), plus a pure streaming kernel with k_propagate's
// byte pattern (read 3 rows, write 4 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../include/cssm_numerics.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_piece(double* __restrict__ out, uint64_t n, uint64_t seed, uint32_t step, const double* __restrict__ gtab) {
  __shared__ double tab[CSSM_TAB_SIZE];
  for (int i = threadIdx.x; i < CSSM_TAB_SIZE; i += 256) tab[i] = gtab[i];
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      cssm_u32x4 b;
      if (MODE == 0) { b.v[0] = (uint32_t)i * 2654435761u + p; b.v[1] = (uint32_t)(i >> 3) ^ 0x9E3779B9u; b.v[2] = b.v[0] ^ step; b.v[3] = b.v[1] + p; }
      else b = cssm_philox_draw(seed, i, step, 0, p);
      double u1 = cssm_u01_open0(b.v[0], b.v[1]), u2 = cssm_u01(b.v[2], b.v[3]);
      if (MODE <= 1) { acc += u1 + u2; continue; }
      double l = cssm_log_unit(u1, tab);
      if (MODE == 2) { acc += l + u2; continue; }
      double r = cssm_sqrt(-2.0 * l);
      if (MODE == 3) { acc += r + u2; continue; }
      double sn, cs;
      cssm_sincos2pi(u2, &sn, &cs);
      if (MODE == 4) { acc += r * cs + r * sn; continue; }
      acc += cssm_exp(r * cs) + r * sn;   // MODE 5: + one exp
    }
    out[i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ src, double* __restrict__ dst, size_t stride, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    double a = src[i], b = src[stride + i], c = src[2 * stride + i];
    dst[i] = a + 1.0; dst[stride + i] = b + 1.0; dst[2 * stride + i] = c + 1.0; dst[3 * stride + i] = a + b + c;
  }
}

template <typename F>
static float time_it(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main() {
  const uint64_t sizes[] = {1ull << 20, 1ull << 22, 1ull << 24};
  for (uint64_t n : sizes) {
    double *out, *src, *dst, *gtab;
    CHECK(hipMalloc(&gtab, sizeof(CSSM_TAB))); CHECK(hipMemcpy(gtab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&src, n * 8 * 3)); CHECK(hipMalloc(&dst, n * 8 * 4));
    CHECK(hipMemset(src, 0, n * 8 * 3));
    for (int grid : {1024, 2048, 4096, 8192}) {
      if ((uint64_t)grid * 256 > n) continue;
      printf("N=%llu grid=%d:", (unsigned long long)n, grid);
      printf(" nophilox %.1f", time_it([&] { k_piece<0><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" philox %.1f", time_it([&] { k_piece<1><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" +log %.1f", time_it([&] { k_piece<2><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" +sqrt %.1f", time_it([&] { k_piece<3><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" +sincos %.1f", time_it([&] { k_piece<4><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" +exp %.1f", time_it([&] { k_piece<5><<<grid, 256>>>(out, n, 1, 2, gtab); }, 20));
      printf(" | stream(56B) %.1f us\n", time_it([&] { k_stream<<<grid, 256>>>(src, dst, n, n); }, 20));
    }
    hipFree(out); hipFree(src); hipFree(dst);
  }
  return 0;
}

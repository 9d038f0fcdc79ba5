// What do agent-scope (sc1) accesses cost for BULK data on gfx950?  A persistent series kernel (LABNOTES_rounds1-3.md, section 5b) would
// have to move the particle state with them, since nothing else keeps the 8 L2s coherent inside a kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/sc1_stream_bench.hip -o tools/sc1_stream_bench.bin && tools/sc1_stream_bench.bin
// Pattern of k_propagate at d = 3: read 3 rows, write 3 rows + 1 row of doubles (56 B per particle), a few flops between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool SC1> __device__ __forceinline__ double ld(const double* p) {
  if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool SC1> __device__ __forceinline__ void st(double* p, double v) {
  if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool SC1>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ src, double* __restrict__ dst, double* __restrict__ w, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double a = ld<SC1>(src + i), b = ld<SC1>(src + n + i), c = ld<SC1>(src + 2 * n + i);
    st<SC1>(dst + i, a * 0.5 + 1.0); st<SC1>(dst + n + i, b * 0.5 + a); st<SC1>(dst + 2 * n + i, c * 0.5 + b);
    st<SC1>(w + i, a + b + c);
  }
}

int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t n : {(size_t)1 << 20, (size_t)1 << 24}) {
    double *src, *dst, *w;
    CK(hipMalloc(&src, 3 * n * 8)); CK(hipMalloc(&dst, 3 * n * 8)); CK(hipMalloc(&w, n * 8));
    CK(hipMemset(src, 0, 3 * n * 8));
    const int grid = (int)((n / 256 < 8192) ? n / 256 : 8192);
    for (int sc1 = 0; sc1 < 2; ++sc1) {
      float best = 1e9f;
      for (int rep = 0; rep < 30; ++rep) {
        CK(hipEventRecord(e0));
        if (sc1) hipLaunchKernelGGL(k_stream<true>, dim3(grid), dim3(256), 0, 0, src, dst, w, n);
        else hipLaunchKernelGGL(k_stream<false>, dim3(grid), dim3(256), 0, 0, src, dst, w, n);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 5 && ms < best) best = ms;
      }
      printf("N = %zu, %s accesses: %.1f us, %.2f TB/s of the 56 B/particle\n", n, sc1 ? "agent-scope (sc1)" : "plain", best * 1e3f,
             56.0 * n / (best * 1e-3) / 1e12);
    }
    CK(hipFree(src)); CK(hipFree(dst)); CK(hipFree(w));
  }
  return 0;
}

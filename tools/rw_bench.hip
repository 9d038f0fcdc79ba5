// rw_bench.hip -- what a kernel of k_offspring's traffic shape costs with nothing else in it: every lane reads 4 consecutive doubles (two
// 16-byte loads) and writes 4 consecutive 32-bit words (one 16-byte store, plain or write-through), N = 2^24 and 2^20.
//   hipcc -O3 --offload-arch=gfx950 tools/rw_bench.hip -o tools/rw_bench.bin && tools/rw_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int SC1, int CHUNKS>
__global__ __launch_bounds__(256) void k_rw(const double* __restrict__ w, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t q = CHUNKS * 256u;
  const uint32_t w_lo = blockIdx.x * 4u * q + wid * q;
  for (uint32_t c = 0; c < CHUNKS; ++c) {
    const uint32_t i0 = w_lo + c * 256u + lane * 4u;
    if (i0 + 4 > n) break;
    const double2 a = *reinterpret_cast<const double2*>(w + i0);
    const double2 b = *reinterpret_cast<const double2*>(w + i0 + 2);
    u32x4 v; v.x = (uint32_t)(a.x * 3.0); v.y = (uint32_t)(a.y * 3.0); v.z = (uint32_t)(b.x * 3.0); v.w = (uint32_t)(b.y * 3.0);
    uint32_t* p = out + i0;
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(p) = v;
  }
}
template <int SC1, int CHUNKS> float run(const double* w, uint32_t* out, uint32_t n, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)((n + CHUNKS * 1024 - 1) / (CHUNKS * 1024));
  k_rw<SC1, CHUNKS><<<grid, 256>>>(w, out, n);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k_rw<SC1, CHUNKS><<<grid, 256>>>(w, out, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  for (uint32_t n : {1u << 20, 1u << 24}) {
    double* w; uint32_t* out; double* big;
    hipMalloc(&w, (size_t)n * 8); hipMalloc(&out, (size_t)n * 4); hipMalloc(&big, 1u << 30);
    hipMemset(w, 0, (size_t)n * 8); hipMemset(big, 1, 1u << 30);
    printf("N = %u: 12 bytes per particle = %.1f MB\n", n, n * 12.0 / 1e6);
    printf("  plain  stores, 1 chunk per wave : %7.2f us\n", run<0, 1>(w, out, n, 50));
    printf("  sc1    stores, 1 chunk per wave : %7.2f us\n", run<1, 1>(w, out, n, 50));
    printf("  plain  stores, 4 chunks per wave: %7.2f us\n", run<0, 4>(w, out, n, 50));
    printf("  sc1    stores, 4 chunks per wave: %7.2f us\n", run<1, 4>(w, out, n, 50));
    hipFree(w); hipFree(out); hipFree(big);
  }
  return 0;
}

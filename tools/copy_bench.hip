// What does a plain device-to-device copy reach on this GPU?  (the "copy ceiling" bench.py reports beside the 8 TB/s spec)
//   hipcc -O3 --offload-arch=gfx950 tools/copy_bench.hip -o tools/copy_bench.bin && tools/copy_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
  }
  for (; i < n16; i += stride) d[i] = s[i];
}
template <int U, bool NT>
static double run(const v4u* a, v4u* b, size_t n16, int grid, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k_copy<U, NT><<<grid, 256>>>(a, b, n16); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k_copy<U, NT><<<grid, 256>>>((r & 1) ? b : a, (r & 1) ? (v4u*)a : b, n16);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return 2.0 * n16 * 16 * reps / (ms * 1e-3) / 1e9;
}
int main() {
  for (size_t bytes : {(size_t)1 << 28, (size_t)1 << 30}) {
    v4u *a, *b; const size_t n16 = bytes / 16;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    for (int grid : {2048, 4096, 8192, 16384, 65536}) {
      printf("bytes %zu MiB grid %5d: U1 %.0f  U2 %.0f  U4 %.0f  U4nt %.0f  U8 %.0f  U8nt %.0f GB/s\n", bytes >> 20, grid,
             run<1, false>(a, b, n16, grid, 10), run<2, false>(a, b, n16, grid, 10), run<4, false>(a, b, n16, grid, 10),
             run<4, true>(a, b, n16, grid, 10), run<8, false>(a, b, n16, grid, 10), run<8, true>(a, b, n16, grid, 10));
    }
    float ms; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int r = 0; r < 10; ++r) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("bytes %zu MiB hipMemcpyDtoD: %.0f GB/s\n", bytes >> 20, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    (void)hipFree(a); (void)hipFree(b);
  }
  return 0;
}

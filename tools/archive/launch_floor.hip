// launch_floor.hip -- what a dependent kernel launch costs on this GPU, and what each dependent memory hop inside it adds.
// build: hipcc -O3 --offload-arch=gfx950 -o launch_floor tools/launch_floor.hip ; run: ./launch_floor
// A chain of K launches on one stream (each waits for the one before); the kernel chases `hops` pointers (every hop a
// load whose address is the previous load's value, the data written by the PREVIOUS launch, so nothing is cached across
// launches) and then does `valu` dependent fp64 FMAs per thread.  Prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_chain(unsigned* __restrict__ buf, int hops, int valu, double* out, int grid_stride) {
  unsigned p = blockIdx.x * 64u;
  for (int h = 0; h < hops; ++h) p = buf[p + (threadIdx.x & 63)] ;          // dependent loads
  double a = (double)p * 1e-9 + 1.0;
  for (int i = 0; i < valu; ++i) a = a * 1.0000001 + 1e-7;                  // dependent VALU chain
  if (threadIdx.x == 0) { buf[(blockIdx.x * 64u + 4096u * (unsigned)(hops & 1)) % 4096u] = (unsigned)(blockIdx.x * 64u); }
  if (a == 12345.678) out[0] = a;
}
int main() {
  unsigned* buf; double* out;
  hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64);
  std::vector<unsigned> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)((i / 64 * 64 + 64 * 17) % 4096);
  hipMemcpy(buf, h.data(), 1 << 20, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int K = 2000;
  const int cfg[][4] = {{1, 64, 0, 0}, {1, 256, 0, 0}, {1, 256, 1, 0}, {1, 256, 2, 0}, {1, 256, 4, 0}, {1, 256, 8, 0},
                        {1, 256, 0, 1000}, {1, 256, 0, 4000}, {256, 256, 0, 0}, {256, 256, 4, 0}, {1024, 256, 4, 1000}, {1024, 256, 0, 0}};
  for (auto& c : cfg) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_chain, dim3(c[0]), dim3(c[1]), 0, st, buf, c[2], c[3], out, 0);
    hipEventRecord(e0, st);
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_chain, dim3(c[0]), dim3(c[1]), 0, st, buf, c[2], c[3], out, 0);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d block %3d hops %d valu %4d : %6.2f us per launch\n", c[0], c[1], c[2], c[3], ms * 1e3 / K);
  }
  return 0;
}

// Cost of a device-scope grid barrier between co-resident blocks on gfx950 (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_bench.hip -o tools/barrier_bench.bin && tools/barrier_bench.bin
// Each round: every block writes one value, barrier, reads the value of the block "opposite" to it (another XCD) and
// checks it -- so the figure includes the release / acquire traffic a real stage boundary needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifdef NOFENCE   // exchanged data travels with agent-scope (sc1) accesses: no L2 write-back / invalidate at the barrier
#define FENCE() __builtin_amdgcn_s_waitcnt(0)
#define PUT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GET(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FENCE() __threadfence()
#define PUT(p, v) (*(p) = (v))
#define GET(p) __builtin_nontemporal_load(p)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
  __shared__ int s_abort;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_abort = 0;
    FENCE();                                           // release: this block's writes are visible device-wide
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { s_abort = 1; break; }  // exit condition: never hang the box
    }
    FENCE();                                           // acquire
  }
  __syncthreads();
  return s_abort == 0;
}

// Two-level barrier: blocks arrive on the counter of their group (own 128-byte line), the last arrival of a group on the
// top counter, the last group publishes the round in a flag every block spins on.  Counters only grow (no reset races).
__device__ __forceinline__ bool tree_barrier(unsigned* base, unsigned round, unsigned gsize) {
  __shared__ int s_abort2;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_abort2 = 0;
    const unsigned nb = gridDim.x, g = blockIdx.x / gsize, ng = (nb + gsize - 1) / gsize;
    const unsigned in_g = (g + 1 == ng) ? nb - g * gsize : gsize;
    unsigned* flag = base; unsigned* top = base + 32; unsigned* cnt = base + 64 + g * 32;
    FENCE();
    const unsigned a = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == in_g * (round + 1) - 1) {
      const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == ng * (round + 1) - 1) __hip_atomic_store(flag, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { s_abort2 = 1; break; }
    }
    FENCE();
  }
  __syncthreads();
  return s_abort2 == 0;
}

__global__ __launch_bounds__(256) void k_rounds(unsigned* counter, unsigned* data, unsigned rounds, unsigned* bad, int payload, unsigned gsize) {
  const unsigned nb = gridDim.x;
  for (unsigned r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < payload; i += blockDim.x) PUT(&data[(size_t)blockIdx.x * payload + i], r * 7919u + blockIdx.x + i);
    if (!(gsize ? tree_barrier(counter, r, gsize) : grid_barrier(counter, nb * (r + 1)))) { if (threadIdx.x == 0) atomicAdd(bad, 1u << 20); return; }
    const unsigned peer = (blockIdx.x + nb / 2 + 1) % nb;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) {
      const unsigned v = GET(&data[(size_t)peer * payload + i]);
      if (v != r * 7919u + peer + i) atomicAdd(bad, 1u);
    }
    // (no second barrier: the next round's writes go to the block's own slots, read by others only after the next barrier;
    //  a reader still reading round r while the owner writes round r + 1 would be a race -- use two buffers)
    data += (size_t)nb * payload * ((r & 1) ? -1 : 1);
  }
}

int main() {
  unsigned *counter, *data, *bad;
  const int payload = 256;
  CK(hipMalloc(&counter, 4 * (64 + 1024 * 32))); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&data, 2 * 1024 * payload * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (unsigned gsize : {0u, 8u, 16u, 32u})
  for (int blocks : {64, 128, 256, 512, 1024}) {
    for (unsigned rounds : {1u, 1001u}) {
      CK(hipMemset(counter, 0, 4 * (64 + 1024 * 32))); CK(hipMemset(bad, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rounds, dim3(blocks), dim3(256), 0, 0, counter, data, rounds, bad, payload, gsize);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      static float base = 0;
      if (rounds == 1u) base = ms;
      else printf("group %2u blocks %4d: %.2f us per barrier round (1000 rounds, launch %.1f us), mismatches %u\n", gsize, blocks, (ms - base) * 1e3f / 1000.f, base * 1e3f, hb);
    }
  }
  return 0;
}

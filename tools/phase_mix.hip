// phase_mix.hip -- static instruction counts of the PHASES of the fused kernel, each compiled as a kernel of its own against the contract
// header (tools/phase_mix.py reads the assembly): which class of instructions the per-particle arithmetic of k_propagate<1, ...> is made of.
//   hipcc -O3 -std=c++17 -ffp-contract=off -mfma --offload-arch=gfx950 -S -I include -I composablestatespacemodels_amd/csrc tools/phase_mix.hip
#include "cssm_device.hip.h"
#define K(name) extern "C" __global__ void name(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ tabg, uint64_t seed)
// every kernel: one thread, its inputs from memory, its outputs to memory; `base` = the same skeleton with nothing in it
K(ph_base) { const double x = in[threadIdx.x]; out[threadIdx.x] = x; }
K(ph_philox) {   // one Philox4x32-7 block: two Box-Muller pairs = four normals = the variates of FOUR particle-steps at d = 1
  const cssm_u32x4 b = cssm_philox_draw(seed, threadIdx.x, (uint32_t)in[0], CSSM_STREAM_STEP, 0u);
  out[threadIdx.x] = (double)(b.v[0] ^ b.v[1] ^ b.v[2] ^ b.v[3]);
}
K(ph_boxmuller) {   // one pair: log of the radius, square root, table sine / cosine = the normals of TWO particle-steps at d = 1
  __shared__ double tab[CSSM_TAB_SIZE];
  for (int i = threadIdx.x; i < CSSM_TAB_SIZE; i += blockDim.x) tab[i] = tabg[i];
  __syncthreads();
  const uint64_t u = cssm_d2u(in[threadIdx.x]);
  double z0, z1;
  cssm_normal_pair64((uint32_t)u, (uint32_t)(u >> 32), tab, &z0, &z1);
  out[2 * threadIdx.x] = z0; out[2 * threadIdx.x + 1] = z1;
}
K(ph_exp) { out[threadIdx.x] = cssm_exp(in[threadIdx.x]); }                      // lambda = exp(gamma) of the Poisson density
K(ph_exp_le0) { out[threadIdx.x] = cssm_exp_le0(cssm_min_c(in[threadIdx.x], CSSM_REF_BELOW)); }   // w1 = exp(min(w - c, 2^-20))
K(ph_fix) {   // the weight on the 2^-96 grid + the 128-bit accumulation
  cssm_u128 a = cssm_fix_from_unit(in[threadIdx.x]);
  a = cssm_u128_add(a, cssm_fix_from_unit(in[threadIdx.x + 64]));
  out[2 * threadIdx.x] = cssm_u2d(a.lo); out[2 * threadIdx.x + 1] = cssm_u2d(a.hi);
}
K(ph_wavesum) {   // the block's end: one 128-bit wave reduction per thread-range (amortised over the range's particles)
  cssm_u128 a; a.lo = cssm_d2u(in[threadIdx.x]); a.hi = cssm_d2u(in[threadIdx.x + 64]);
  a = wave_sum_u128(a);
  out[0] = cssm_u2d(a.lo ^ a.hi);
}

// cssm_residual.hip -- residual resampling in its DOCUMENTED intent, offered as an extension of the Resample[A] seam.
//
// The reference's Resampling.residualResampling (model/Resampling.scala:124-146) cannot run as written: it hands
// `Vector.range(1, m)` (m - 1 items) with n weights to multinomialResampling, indexes the particles with what comes back and
// exp-normalises weights that stepFilter has already exponentiated.  Its scaladoc states what it means to do -- "particle (xi, wi)
// appears ki = n * wi times; resample m = n - total allocated particles according to w = n * wi - ki using other resampling
// technique" -- and that is what is built here, labelled as an extension (cssm_resample_residual; the reference's own three
// resamplers are cssm_resample):
//
//   p_i = RN(RN(F_i) / RN(S)) * n          F_i = floor(w_i 2^96), S = sum F_i (exact integer sums, numerics contract section 2)
//   k_i = floor(p_i)                       copies of particle i, slots [K_i, K_i + k_i), K_i = k_0 + .. + k_{i-1} (particle order)
//   r_i = p_i - k_i                        residual weight (exact subtraction), m = n - sum k_i further slots
//   slot sum k + t (t < m) <- the first j with C_j >= u_t,   C_j = RN(RN(R_j) / RN(R)), R_j = sum_{i<=j} floor(r_i 2^96),
//                                          u_t = cssm_multi_uniform(seed, step, t)   (the multinomial resampler's draws and search)
//
// Every operation is an IEEE operation or an exact integer sum: oracle/cssm_oracle.c (oracle_resample_residual) restates it on the
// CPU and the two agree bit for bit (tests/test_gpu_parity.py::test_residual_resampling_extension_matches_oracle).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "cssm_internal.h"
#include "cssm_kernels.hip.h"

// k_i and r_i of every particle; per block the sum of its k_i
static __global__ __launch_bounds__(CSSM_BLOCK) void k_residual_split(const double* __restrict__ w, uint64_t n, const Scalars* __restrict__ sc,
                                                               uint32_t* __restrict__ k_out, double* __restrict__ r_out, uint32_t* __restrict__ block_k) {
  __shared__ uint32_t s_k[CSSM_BLOCK / 64];
  const double totd = cssm_u128_to_double(sc->S_tot), nd = (double)n;
  const uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x;
  uint32_t k = 0u;
  if (i < n) {
    const double p = (cssm_u128_to_double(cssm_fix_from_double(w[i])) / totd) * nd;
    k = (p >= nd) ? (uint32_t)n : (uint32_t)p;
    k_out[i] = k;
    r_out[i] = p - (double)k;
  }
  uint32_t v = k;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t t = 0; for (int q = 0; q < CSSM_BLOCK / 64; ++q) t += s_k[q]; block_k[blockIdx.x] = t; }
}
// exclusive prefix of the blocks' sums (one block; u64 running sum: the total must not exceed n, checked on the host)
static __global__ void k_residual_scan_blocks(uint32_t* __restrict__ block_k, uint32_t nblocks, unsigned long long* __restrict__ total) {
  __shared__ unsigned long long s_part[1024];
  const uint32_t per = (nblocks + 1023u) / 1024u, b0 = threadIdx.x * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
  unsigned long long a = 0;
  for (uint32_t b = b0; b < b1; ++b) a += block_k[b];
  s_part[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int q = 0; q < 1024; ++q) { const unsigned long long v = s_part[q]; s_part[q] = run; run += v; }
    *total = run;
  }
  __syncthreads();
  unsigned long long run = s_part[threadIdx.x];
  for (uint32_t b = b0; b < b1; ++b) { const uint32_t v = block_k[b]; block_k[b] = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run); run += v; }
}
// k_i -> INCLUSIVE prefix K_i + k_i in place (block-local scan on top of the block's offset)
static __global__ __launch_bounds__(CSSM_BLOCK) void k_residual_prefix(uint32_t* __restrict__ k_io, uint64_t n, const uint32_t* __restrict__ block_k) {
  __shared__ uint32_t s_w[CSSM_BLOCK / 64];
  const uint64_t i = (uint64_t)blockIdx.x * CSSM_BLOCK + threadIdx.x;
  uint32_t v = (i < n) ? k_io[i] : 0u;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
  if (lane == 63) s_w[wid] = v;
  __syncthreads();
  uint32_t base = block_k[blockIdx.x];
  for (int q = 0; q < wid; ++q) base += s_w[q];
  if (i < n) k_io[i] = base + v;
}
// the deterministic slots: slot s < K belongs to the first particle whose inclusive prefix exceeds s
static __global__ void k_residual_fill(const uint32_t* __restrict__ inc, uint64_t n, uint64_t K, uint32_t* __restrict__ anc) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < K; s += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t lo = 0, hi = n - 1;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)inc[mid] > s) hi = mid; else lo = mid + 1; }
    anc[s] = (uint32_t)lo;
  }
}
// the m residual draws: draw t takes the first j with C_j >= u_t (k_multinomial's search), into slot K + t
static __global__ void k_residual_draws(const double* __restrict__ cum, uint64_t n, uint64_t seed, uint32_t step, uint64_t K, uint64_t m, uint32_t* __restrict__ anc) {
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (uint64_t)gridDim.x * blockDim.x) {
    const double ut = cssm_multi_uniform(seed, step, t);
    uint64_t lo = 0, hi = n - 1;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (cum[mid] >= ut) hi = mid; else lo = mid + 1; }
    anc[K + t] = (uint32_t)lo;
  }
}

extern "C" int cssm_resample_residual(const double* w, size_t n, uint64_t seed, uint32_t step, uint32_t* anc, int device) {
  if (!w || !anc) return fail(CSSM_EINVAL_ARG, "null argument");
  if (n < 1 || n >= 0xffffffffull) return fail(CSSM_EINVAL_ARG, "n out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(CSSM_EHIP, "no HIP device available (this library has no CPU path)");
  HIP_TRY(hipSetDevice(device));
  const uint32_t ntiles = (uint32_t)((n + CSSM_TILE - 1) / CSSM_TILE);
  const uint32_t sup = (ntiles + 1023u) / 1024u, nunits = (ntiles + sup - 1) / sup;
  const size_t stride = (size_t)ntiles * CSSM_TILE;
  const uint32_t nblocks = (uint32_t)((n + CSSM_BLOCK - 1) / CSSM_BLOCK);
  double *d_w = nullptr, *d_r = nullptr, *d_cum = nullptr, *d_tab = nullptr; uint32_t *d_k = nullptr, *d_bk = nullptr, *d_anc = nullptr, *d_end = nullptr;
  cssm_u128 *tS = nullptr, *tS2 = nullptr, *tP = nullptr; Scalars* sc = nullptr; StepRec* d_rec = nullptr; unsigned long long* d_total = nullptr;
  hipStream_t st = nullptr;
  int rc = CSSM_OK;
  StepRec hrec; memset(&hrec, 0, sizeof hrec); hrec.step = step;
  Scalars hs;
  unsigned long long K = 0;
  std::vector<double> wscaled;
#define RS_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(CSSM_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); goto done; } } while (0)
  RS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  RS_TRY(hipMalloc(&d_w, stride * 8)); RS_TRY(hipMalloc(&d_r, stride * 8)); RS_TRY(hipMalloc(&d_cum, stride * 8));
  RS_TRY(hipMalloc(&d_k, stride * 4)); RS_TRY(hipMalloc(&d_bk, (size_t)nblocks * 4)); RS_TRY(hipMalloc(&d_anc, stride * 4)); RS_TRY(hipMalloc(&d_end, stride * 4));
  RS_TRY(hipMalloc(&tS, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tS2, ntiles * sizeof(cssm_u128))); RS_TRY(hipMalloc(&tP, ntiles * sizeof(cssm_u128)));
  RS_TRY(hipMalloc(&sc, sizeof(Scalars))); RS_TRY(hipMalloc(&d_rec, sizeof(StepRec))); RS_TRY(hipMalloc(&d_tab, sizeof(CSSM_TAB))); RS_TRY(hipMalloc(&d_total, 8));
  RS_TRY(hipMemcpyAsync(d_tab, CSSM_TAB, sizeof(CSSM_TAB), hipMemcpyHostToDevice, st));
  RS_TRY(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
  RS_TRY(hipMemsetAsync(d_r, 0, stride * 8, st));
  {   // (weights of any scale: brought up by an exact power of two as cssm_resample does -- the sums live on a 2^-96 grid)
    double wmax = 0.0;
    for (size_t i = 0; i < n; ++i) if (w[i] > wmax) wmax = w[i];
    if (wmax > 0.0 && wmax < 0x1.0p-32) {
      int e = 0;
      (void)std::frexp(wmax, &e);
      wscaled.resize(n);
      for (size_t i = 0; i < n; ++i) wscaled[i] = std::ldexp(w[i], -e);
      w = wscaled.data();
    }
  }
  RS_TRY(hipMemcpyAsync(d_w, w, n * 8, hipMemcpyHostToDevice, st));
  RS_TRY(hipMemcpyAsync(d_rec, &hrec, sizeof hrec, hipMemcpyHostToDevice, st));
  // S = sum F_i
  hipLaunchKernelGGL(k_tile_sums, dim3((int)nunits), dim3(CSSM_BLOCK), 0, st, d_w, (uint64_t)n, sc, tS, tS2, ntiles, sup, nunits, 1, -1, (const double*)nullptr, d_tab,
                     (const StepRec*)d_rec, 0u);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tS, tS2, tP, nunits, sc, (uint64_t)n, 1, (double*)nullptr, (int32_t*)nullptr, 0u,
                     (const double*)nullptr, (unsigned long long*)nullptr, 0, 0u, 1);
  // k_i, r_i, the copies' slots
  hipLaunchKernelGGL(k_residual_split, dim3(nblocks), dim3(CSSM_BLOCK), 0, st, (const double*)d_w, (uint64_t)n, (const Scalars*)sc, d_k, d_r, d_bk);
  hipLaunchKernelGGL(k_residual_scan_blocks, dim3(1), dim3(1024), 0, st, d_bk, nblocks, d_total);
  hipLaunchKernelGGL(k_residual_prefix, dim3(nblocks), dim3(CSSM_BLOCK), 0, st, d_k, (uint64_t)n, (const uint32_t*)d_bk);
  RS_TRY(hipGetLastError());
  RS_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&hs, sc), hipMemcpyDeviceToHost, st));
  RS_TRY(hipMemcpyAsync(&K, d_total, 8, hipMemcpyDeviceToHost, st));
  RS_TRY(hipStreamSynchronize(st));
  if (hs.S_tot.lo == 0 && hs.S_tot.hi == 0) { rc = fail(CSSM_ENONFINITE, "all weights are zero (the reference divides by a zero total)"); goto done; }
  if (K > (unsigned long long)n) { rc = fail(CSSM_ESTATE, "residual resampling: %llu deterministic copies for %zu slots", K, n); goto done; }
  if (K > 0) hipLaunchKernelGGL(k_residual_fill, dim3(grid_for(K, 256, kGridCap)), dim3(256), 0, st, (const uint32_t*)d_k, (uint64_t)n, (uint64_t)K, d_anc);
  if (K < (unsigned long long)n) {
    // the residual weights' cumulative distribution (the multinomial resampler's: contract sums of r), then m draws
    const uint64_t m = (uint64_t)n - K;
    RS_TRY(hipMemsetAsync(sc, 0, sizeof(Scalars), st));
    hipLaunchKernelGGL(k_tile_sums, dim3((int)nunits), dim3(CSSM_BLOCK), 0, st, d_r, (uint64_t)n, sc, tS, tS2, ntiles, sup, nunits, 1, -1, (const double*)nullptr, d_tab,
                       (const StepRec*)d_rec, 0u);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tS, tS2, tP, nunits, sc, (uint64_t)n, 1, (double*)nullptr, (int32_t*)nullptr, 0u,
                       (const double*)nullptr, (unsigned long long*)nullptr, 0, 0u, 1);
    hipLaunchKernelGGL((k_offspring<true, false, CSSM_RESAMPLE_MULTINOMIAL>), dim3((int)nunits), dim3(CSSM_BLOCK), 0, st,
                       d_r, (uint64_t)n, sc, (const cssm_u128*)tP, (const cssm_u128*)tS2, d_rec, (uint64_t)n, d_end, d_end /* (unused: no runs are written) */, ntiles, sup, nunits, 1, 0,
                       (double*)nullptr, (int32_t*)nullptr, 0u, 0, (const unsigned long long*)nullptr, 0, 1, 1, seed, d_cum, d_tab, 0,
                       (unsigned long long*)nullptr, 0u, (uint32_t)n, 5u);
    RS_TRY(hipGetLastError());
    RS_TRY(hipMemcpyAsync(CSSM_SC_TAIL_ARGS(&hs, sc), hipMemcpyDeviceToHost, st));
    RS_TRY(hipStreamSynchronize(st));
    if (hs.S_tot.lo == 0 && hs.S_tot.hi == 0) { rc = fail(CSSM_ENONFINITE, "residual resampling: %llu slots are left and every residual weight is zero", (unsigned long long)m); goto done; }
    hipLaunchKernelGGL(k_residual_draws, dim3(grid_for(m, 256, kGridCap)), dim3(256), 0, st, (const double*)d_cum, (uint64_t)n, seed, step, (uint64_t)K, m, d_anc);
  }
  RS_TRY(hipGetLastError());
  RS_TRY(hipMemcpyAsync(anc, d_anc, n * 4, hipMemcpyDeviceToHost, st));
  RS_TRY(hipStreamSynchronize(st));
done:
#undef RS_TRY
  void* ptrs[] = {d_w, d_r, d_cum, d_tab, d_k, d_bk, d_anc, d_end, tS, tS2, tP, sc, d_rec, d_total};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

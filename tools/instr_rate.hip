// Issue rate of single VALU / LDS instructions on gfx950 (one wave per SIMD and four waves per SIMD).
// Build: hipcc -O2 --offload-arch=gfx950 tools/instr_rate.hip -o /tmp/instr_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP8(x) x x x x x x x x

#define KERNEL(name, decl, body, sink)                                             \
  __global__ void name(double* out, int n) {                                        \
    decl;                                                                           \
    for (int i = 0; i < n; ++i) { REP8(body) }                                      \
    sink;                                                                           \
  }

KERNEL(k_fma_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0000001; double c = 0.5,
       asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_mul_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0000001,
       asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_add_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0000001,
       asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_mad_u64_u32, unsigned long long a0 = threadIdx.x; unsigned long long a1 = a0 + 1; unsigned long long a2 = a0 + 2; unsigned long long a3 = a0 + 3; unsigned b = 0xD2511F53u,
       asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_mul_hi_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xD2511F53u,
       asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_mul_lo_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xD2511F53u,
       asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_mul_u32_u24, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0x511F53u,
       asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_xor_b32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xD2511F53u,
       asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_lshl_b64, unsigned long long a0 = threadIdx.x; unsigned long long a1 = a0 + 1; unsigned long long a2 = a0 + 2; unsigned long long a3 = a0 + 3; unsigned b = 1u,
       asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_lshl_add_u64, unsigned long long a0 = threadIdx.x; unsigned long long a1 = a0 + 1; unsigned long long a2 = a0 + 2; unsigned long long a3 = a0 + 3; unsigned long long b = 12345u,
       asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_cvt_f64_u32, double a0 = 0; double a1 = 0; double a2 = 0; double a3 = 0; unsigned b = threadIdx.x,
       asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_rsq_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0,
       asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_rcp_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0,
       asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_sqrt_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 1.0,
       asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_ldexp_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; int b = 1,
       asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_cndmask, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u,
       asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_fma_f32, float a0 = threadIdx.x; float a1 = a0 + 1; float a2 = a0 + 2; float a3 = a0 + 3; float b = 1.0000001f,
       asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_bpermute, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = ((threadIdx.x + 1) & 63) * 4,
       asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_dpp_mov, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u,
       asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))

KERNEL(k_cndmask_e64, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u; unsigned long long m = 0x5555555555555555ull,
       asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(m));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_cndmask_indep, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u; unsigned c = threadIdx.x * 3,
       asm volatile("v_cndmask_b32 %0, %5, %4, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %5, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_cmp_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 7.0,
       asm volatile("v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_cmp_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u,
       asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_cmp_cnd_pair, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 7u,
       asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_max_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 7.0,
       asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_and_b32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xfff7u,
       asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_add_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xfff7u,
       asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_addc_pair, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 0xfffffff7u,
       asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_bfe_u32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 3u,
       asm volatile("v_bfe_u32 %0, %0, %4, 11\n v_bfe_u32 %1, %1, %4, 11\n v_bfe_u32 %2, %2, %4, 11\n v_bfe_u32 %3, %3, %4, 11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_mov_b32, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 3u,
       asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_mov_b64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 3.0,
       asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_readlane, unsigned a0 = threadIdx.x; unsigned a1 = a0 + 1; unsigned a2 = a0 + 2; unsigned a3 = a0 + 3; unsigned b = 3u,
       asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "s22", "s23");,
       out[threadIdx.x] = (double)(a0 + a1 + a2 + a3))
KERNEL(k_floor_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 3.0,
       asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_cvt_i32_f64, double a0 = threadIdx.x; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; unsigned r0 = 0; unsigned r1 = 0,
       asm volatile("v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %4, %2\n v_cvt_i32_f64 %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(r0), "+v"(r1));,
       out[threadIdx.x] = a0 + a1 + a2 + a3 + r0 + r1)
KERNEL(k_div_scale_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 3.0,
       asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_scale_f64 %2, vcc, %2, %4, %2\n v_div_scale_f64 %3, vcc, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_div_fmas_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 3.0,
       asm volatile("v_div_fmas_f64 %0, %0, %4, %4\n v_div_fmas_f64 %1, %1, %4, %4\n v_div_fmas_f64 %2, %2, %4, %4\n v_div_fmas_f64 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_div_fixup_f64, double a0 = threadIdx.x + 1.0; double a1 = a0 + 1; double a2 = a0 + 2; double a3 = a0 + 3; double b = 3.0,
       asm volatile("v_div_fixup_f64 %0, %0, %4, %4\n v_div_fixup_f64 %1, %1, %4, %4\n v_div_fixup_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)
KERNEL(k_ds_read_b64, double a0 = 0; double a1 = 0; double a2 = 0; double a3 = 0; unsigned b = (threadIdx.x & 127) * 16,
       asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:2056\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       out[threadIdx.x] = a0 + a1 + a2 + a3)

typedef void (*kern_t)(double*, int);
struct K { const char* name; kern_t k; };

int main() {
  double* out; hipMalloc(&out, 4096 * 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double mhz = p.clockRate / 1000.0;
  printf("device %s CUs %d clock %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
  K ks[] = {{"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_mad_u64_u32", k_mad_u64_u32},
            {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_xor_b32", k_xor_b32},
            {"v_lshlrev_b64", k_lshl_b64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cvt_f64_u32", k_cvt_f64_u32},
            {"v_rsq_f64", k_rsq_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_ldexp_f64", k_ldexp_f64},
            {"v_cndmask_b32", k_cndmask}, {"v_fma_f32", k_fma_f32}, {"ds_bpermute_b32", k_bpermute}, {"v_mov_b32_dpp", k_dpp_mov},
            {"v_cndmask_e64(sgpr)", k_cndmask_e64}, {"v_cndmask(indep)", k_cndmask_indep}, {"v_cmp_gt_f64", k_cmp_f64}, {"v_cmp_gt_u32", k_cmp_u32},
            {"cmp+cndmask pair", k_cmp_cnd_pair}, {"v_max_f64", k_max_f64}, {"v_and_b32", k_and_b32}, {"v_add_u32", k_add_u32},
            {"add_co+addc pair", k_addc_pair}, {"v_bfe_u32", k_bfe_u32}, {"v_mov_b32", k_mov_b32}, {"v_mov_b64", k_mov_b64},
            {"v_readlane_b32", k_readlane}, {"v_floor_f64", k_floor_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_div_scale_f64", k_div_scale_f64},
            {"v_div_fmas_f64", k_div_fmas_f64}, {"v_div_fixup_f64", k_div_fixup_f64}, {"ds_read_b64", k_ds_read_b64}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {1, 2, 3, 4, 6, 8}) {
    printf("-- %d wave(s) per SIMD\n", waves);
    for (auto& k : ks) {
      if (waves != 1 && waves != 4 && k.k != (kern_t)k_fma_f64 && k.k != (kern_t)k_xor_b32 && k.k != (kern_t)k_mad_u64_u32) continue;
      const int grid = p.multiProcessorCount * waves;   // blocks of 256 threads = 4 waves = one per SIMD
      hipLaunchKernelGGL(k.k, dim3(grid), dim3(256), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.k, dim3(grid), dim3(256), 0, 0, out, ITER);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_wave = (double)ITER * 8 * 4;
      const double cyc = ms * 1e-3 * mhz * 1e6 / (instr_per_wave * waves);
      printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", k.name, ms, cyc);
    }
  }
  return 0;
}

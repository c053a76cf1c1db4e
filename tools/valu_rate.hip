// Micro-benchmark (diagnostics): SIMD cycles per instruction of the VALU forms the DS-CNN unit is made of, with 1, 2 and 4
// wavefronts per SIMD (eight independent destination registers per kind, so no instruction waits for its predecessor).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define KERNEL_T(NAME, ASM, T)                                                                                   \
    __global__ void NAME(float* out, unsigned long long* cyc, int iters) {                                   \
        T a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;                                            \
        T f[8];                                                                                               \
        for (int i = 0; i < 8; ++i) f[i] = a + i;                                                             \
        __syncthreads();                                                                                      \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (int it = 0; it < iters; ++it) {                                                                  \
            _Pragma("unroll") for (int u = 0; u < 32; ++u) {                                                  \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(f[i]) : "v"(b), "v"(a)); \
            }                                                                                                 \
        }                                                                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
        float s = 0;                                                                                          \
        for (int i = 0; i < 8; ++i) s += (float)f[i];                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                     \
    }
#define KERNEL(NAME, ASM) KERNEL_T(NAME, ASM, float)
#define KERNEL_D(NAME, ASM) KERNEL_T(NAME, ASM, double)
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_fmac_dpp, "v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_mix_f32, "v_fma_mix_f32 %0, %0, %1, %2")
KERNEL(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, 0")
KERNEL(k_mixhi, "v_fma_mixhi_f16 %0, %1, %2, 0")
KERNEL(k_cvt_pk, "v_cvt_pk_f16_f32 %0, %1, %2")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL(k_cndmask64v, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL(k_cndmask32r, "v_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL(k_cmp_cnd32, "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
KERNEL(k_cmp_cnd64, "v_cmp_gt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
KERNEL(k_max_i32_0, "v_max_i32 %0, 0, %0")
KERNEL(k_max_f32_0, "v_max_f32 %0, 0, %0")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %1")
KERNEL(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %1")
KERNEL(k_pack, "v_pack_b32_f16 %0, %0, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 1, 5")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL(k_readfirst, "v_readfirstlane_b32 s20, %0")
KERNEL(k_swap, "v_permlane32_swap_b32 %0, %1")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
KERNEL_D(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_D(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL_D(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_max_e64, "v_max_f32_e64 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_cvt_i, "v_cvt_f32_i32 %0, %1")
KERNEL(k_exp, "v_exp_f32 %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %1")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, 1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL_D(k_fma64, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_log, "v_log_f32 %0, %1")
template <typename K>
void run(K kern, int threads, const char* name) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 16);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0; int n = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < threads / 64; ++w) { m += h[g * 16 + w]; ++n; }
    m /= n;
    const double per_wave = m / (iters * 32.0 * 8), waves_per_simd = threads / 256.0;
    printf("%-22s %d wave(s)/SIMD: %.2f cycles per instr per wave, %.2f SIMD-cycles per instr\n", name, threads / 256, per_wave, per_wave / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
#define RUN(K, N) for (int t : {256, 512, 1024}) run(K, t, N);
int main() {
    RUN(k_fma, "v_fma_f32") RUN(k_fmac, "v_fmac_f32") RUN(k_mul, "v_mul_f32") RUN(k_max, "v_max_f32") RUN(k_mov, "v_mov_b32")
    RUN(k_fmac_dpp, "v_fmac_f32_dpp wave_shr") RUN(k_add_dpp, "v_add_f32_dpp row_shr") RUN(k_mix_f32, "v_fma_mix_f32")
    RUN(k_mixlo, "v_fma_mixlo_f16") RUN(k_mixhi, "v_fma_mixhi_f16") RUN(k_cvt_pk, "v_cvt_pk_f16_f32") RUN(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32")
    RUN(k_cndmask, "v_cndmask_b32 vcc") RUN(k_cndmask64, "v_cndmask_b32_e64 sgpr") RUN(k_cndmask64v, "v_cndmask_b32_e64 vcc") RUN(k_cndmask32r, "v_cndmask_b32_e32 d=s0") RUN(k_cmp_cnd32, "v_cmp+v_cndmask e32 vcc (2)") RUN(k_cmp_cnd64, "v_cmp+v_cndmask e64 sgpr (2)") RUN(k_max_i32, "v_max_i32") RUN(k_max_i32_0, "v_max_i32 0,x") RUN(k_max_f32_0, "v_max_f32 0,x") RUN(k_min_f32, "v_min_f32") RUN(k_sub, "v_sub_f32") RUN(k_or, "v_or_b32") RUN(k_xor, "v_xor_b32") RUN(k_cvt_f32_f16, "v_cvt_f32_f16") RUN(k_cvt_f16_f32, "v_cvt_f16_f32") RUN(k_pack, "v_pack_b32_f16") RUN(k_lshl_add, "v_lshl_add_u32") RUN(k_add3, "v_add3_u32") RUN(k_bfe, "v_bfe_u32") RUN(k_ashr, "v_ashrrev_i32") RUN(k_mul_u24, "v_mul_u32_u24") RUN(k_addc, "v_addc_co_u32 vcc") RUN(k_readlane, "v_readlane_b32") RUN(k_readfirst, "v_readfirstlane_b32") RUN(k_swap, "v_permlane32_swap") RUN(k_bperm, "ds_bpermute+wait") RUN(k_and, "v_and_b32") RUN(k_log, "v_log_f32")
    RUN(k_pk_fma, "v_pk_fma_f32") RUN(k_pk_mul, "v_pk_mul_f32") RUN(k_pk_add, "v_pk_add_f32") RUN(k_add, "v_add_f32") RUN(k_max_e64, "v_max_f32_e64")
    RUN(k_max3, "v_max3_f32") RUN(k_med3, "v_med3_f32") RUN(k_mov_dpp, "v_mov_b32_dpp") RUN(k_cvt_i, "v_cvt_f32_i32") RUN(k_exp, "v_exp_f32")
    RUN(k_rcp, "v_rcp_f32") RUN(k_ldexp, "v_ldexp_f32") RUN(k_lshl, "v_lshlrev_b32") RUN(k_addu, "v_add_u32") RUN(k_mad24, "v_mad_u32_u24")
    RUN(k_perm, "v_perm_b32") RUN(k_cmp, "v_cmp_gt_f32 vcc") RUN(k_fma64, "v_fma_f64")
    return 0;
}

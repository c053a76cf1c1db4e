// Micro-benchmark (diagnostics): SIMD cycles per instruction of the VALU forms the DS-CNN unit is made of, with 1, 2 and 4
// wavefronts per SIMD (eight independent destination registers per kind, so no instruction waits for its predecessor).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void NAME(float* out, unsigned long long* cyc, int iters) {                                   \
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;                                        \
        float f[8];                                                                                           \
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
        for (int i = 0; i < 8; ++i) s += f[i];                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                     \
    }
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
    RUN(k_cndmask, "v_cndmask_b32") RUN(k_and, "v_and_b32") RUN(k_log, "v_log_f32")
    return 0;
}

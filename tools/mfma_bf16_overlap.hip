// Micro-benchmark (diagnostics): bf16 MFMA (32x32x16) issue rate, and how much VALU a co-resident wave still gets.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int OP>  // MODE 0: MFMA waves only, 1: VALU waves only, 2: both; OP 0: v_fma_f32, 1: v_dot2_f32_bf16, 2: v_cvt_pk_bf16_f32
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
    floatx16 acc0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, acc1 = acc0;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_wave) {
        if (MODE != 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc1, 0, 0, 0);
                }
            }
    } else {
        if (MODE != 0)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 64; ++u) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (OP == 0) f[i] = fmaf(f[i], b, a);
                        if (OP == 1) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(b), "s"(0x0000BF80u));
                        if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                        if (OP == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(f[i]) : "s"(0xffff0000u));
                        if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "s"(0x07060302u));
                        if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                        if (OP == 6) asm volatile("v_fmac_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(b));
                        if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                        if (OP == 8) asm volatile("v_max_f32 %0, 0, %0" : "+v"(f[i]));
                        if (OP == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                        if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(f[i]) : "v"(b));
                        if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(a));
                        if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(f[i]) : "s"(0xffff0000u), "v"(b));
                        if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(b));
                        if (OP == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
                        if (OP == 15) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(f[i]) : "v"(b), "s"(0xbf800000u));
                        if (OP == 16) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                    }
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int OP>
void run(const char* name) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * 512);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, OP>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[g * 8 + w];
    m /= grid * 4; v /= grid * 4;
    printf("%-28s MFMA waves: %8.0f cycles (%.1f per MFMA)   VALU waves: %8.0f cycles (%.2f per VALU op)\n", name, m,
           m / (iters * 32.0), v, v / (iters * 64.0 * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>("bf16 32x32x16 MFMA alone");
    run<1, 0>("v_fma_f32 waves alone");
    run<2, 0>("bf16 MFMA + v_fma_f32");
    run<1, 1>("v_dot2_f32_bf16 waves alone");
    run<2, 1>("bf16 MFMA + v_dot2_f32_bf16");
    run<1, 2>("v_cvt_pk_bf16_f32 alone");
    run<2, 2>("bf16 MFMA + v_cvt_pk_bf16");
    run<1, 3>("v_and_b32 alone");
    run<1, 4>("v_perm_b32 alone");
    run<1, 5>("v_sub_f32 alone");
    run<1, 6>("v_fmac_f32_dpp alone");
    run<1, 7>("v_mul_f32 alone");
    run<1, 8>("v_max_f32 alone");
    run<1, 9>("v_add_u32 alone");
    run<1, 10>("v_mov_b32 alone");
    run<1, 11>("v_fma_f32 (asm) alone");
    run<1, 12>("v_and_or_b32 alone");
    run<1, 13>("v_cndmask_b32 alone");
    run<1, 14>("v_exp_f32 alone");
    run<2, 3>("bf16 MFMA + v_and_b32");
    run<2, 4>("bf16 MFMA + v_perm_b32");
    run<2, 6>("bf16 MFMA + v_fmac_dpp");
    run<1, 15>("v_fma_mix_f32 alone");
    run<2, 15>("bf16 MFMA + v_fma_mix_f32");
    run<1, 16>("v_cvt_pkrtz_f16_f32 alone");
    run<2, 16>("bf16 MFMA + v_cvt_pkrtz_f16");
    run<2, 5>("bf16 MFMA + v_sub_f32");
    return 0;
}

// Micro-benchmark (diagnostics): does the f32 MFMA overlap with VALU work of a co-resident wavefront?
// 512-thread workgroups (2 waves per SIMD): waves 0-3 run an MFMA loop, waves 4-7 a v_fma loop (or idle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE, bool SMALL>  // 0: MFMA waves only, 1: VALU waves only, 2: both
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    floatx16 acc0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, acc1 = acc0;
    floatx4 s0 = {0,0,0,0}, s1 = s0, s2 = s0, s3 = s0;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_wave) {
        if (MODE != 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (SMALL) {
                        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, s0, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, s1, 0, 0, 0);
                        s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, s2, 0, 0, 0);
                        s3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, s3, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
                    }
                }
            }
    } else {
        if (MODE != 0)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 64; ++u) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], b, a);
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int r = 0; r < 4; ++r) s += s0[r] + s1[r] + s2[r] + s3[r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool SMALL>
void run(const char* name) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * 512);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, SMALL>), dim3(grid), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[g * 8 + w];
    m /= grid * 4; v /= grid * 4;
    printf("%-26s MFMA waves: %8.0f cycles (%.1f per MFMA)   VALU waves: %8.0f cycles (%.2f per v_fma)\n", name, m,
           m / (iters * (SMALL ? 64.0 : 32.0)), v, v / (iters * 64.0 * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, false>("32x32x2 MFMA waves alone");
    run<1, false>("VALU waves alone");
    run<2, false>("32x32x2 + VALU, same SIMDs");
    run<0, true>("16x16x4 MFMA waves alone");
    run<2, true>("16x16x4 + VALU, same SIMDs");
    return 0;
}

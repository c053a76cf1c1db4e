// Micro-benchmark (diagnostics): issue rate of the f32 MFMAs with 1 or 2 wavefronts per SIMD and 1, 2 or 4
// independent accumulators.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SMALL>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    floatx16 acc[4];
    floatx4 acs[4];
    for (int i = 0; i < 4; ++i) { acc[i] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; acs[i] = {0,0,0,0}; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SMALL) acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acs[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, bool SMALL>
void run(int threads, const char* name) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, SMALL>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * (threads / 64));
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double mfma_per_wave = (double)iters * 16 * NACC;
    const double waves_per_simd = threads / 64 / 4.0;
    printf("%-10s threads=%4d acc=%d: %.1f cycles per MFMA per wave, %.1f SIMD-cycles per MFMA\n", name, threads, NACC,
           avg / mfma_per_wave, avg / (mfma_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, false>(256, "32x32x2"); run<2, false>(256, "32x32x2"); run<4, false>(256, "32x32x2");
    run<1, false>(512, "32x32x2"); run<2, false>(512, "32x32x2"); run<4, false>(512, "32x32x2");
    run<2, false>(768, "32x32x2"); run<2, false>(1024, "32x32x2"); run<1, false>(1024, "32x32x2");
    run<1, true>(256, "16x16x4"); run<2, true>(256, "16x16x4"); run<4, true>(256, "16x16x4");
    run<2, true>(512, "16x16x4"); run<4, true>(512, "16x16x4"); run<4, true>(768, "16x16x4"); run<4, true>(1024, "16x16x4");
    return 0;
}

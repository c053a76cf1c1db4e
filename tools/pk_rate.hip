// Micro-benchmark (diagnostics): issue cost of v_fma_f32 vs v_pk_fma_f32 (1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f2 f[8];
    for (int i = 0; i < 8; ++i) f[i] = f2{a + i, a - i};
    const f2 bb = {b, b}, aa = {a, a};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(bb), "v"(aa));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i].x) : "v"(b), "v"(a));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i].x + f[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <bool PK>
void run(int threads, const char* name) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<PK>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 16);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0; int n = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < threads / 64; ++w) { m += h[g * 16 + w]; ++n; }
    m /= n;
    const double per_wave = m / (iters * 32.0 * 8), waves_per_simd = threads / 256.0;
    printf("%-14s threads=%4d: %.2f cycles per instr per wave, %.2f SIMD-cycles per instr\n", name, threads, per_wave, per_wave / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<false>(256, "v_fma_f32"); run<true>(256, "v_pk_fma_f32");
    run<false>(512, "v_fma_f32"); run<true>(512, "v_pk_fma_f32");
    run<false>(1024, "v_fma_f32"); run<true>(1024, "v_pk_fma_f32");
    return 0;
}

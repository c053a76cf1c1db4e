// Micro-benchmark (diagnostics): does a wavefront's own VALU work run under its v_mfma_f32_32x32x16_f16?  One iteration = one MFMA
// (two accumulators alternating) followed by K independent v_fma_f32; 1 and 2 wavefronts per SIMD.  Perfect overlap: max(32, K x cadence)
// cycles per iteration; none: the sum.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_same_wave.hip -o tools/bin/mfma_same_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
template <int K, bool MFMA>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    halfx8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(a + i); bv[i] = (_Float16)(b - i); }
    floatx16 acc0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, acc1 = acc0;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 7]) : "v"(b), "v"(a));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int K, bool MFMA>
void run(int threads) {
    const int grid = 256, iters = 64;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * grid * threads);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * grid * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<K, MFMA>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 16);
    (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0; int n = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < threads / 64; ++w) { m += h[g * 16 + w]; ++n; }
    m /= n;
    printf("K=%2d VALU per %s, %d wave(s)/SIMD: %.1f cycles per iteration per wave\n", K, MFMA ? "MFMA" : "none", threads / 256, m / (iters * 8.0));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int t : {256, 512}) {
        if (t == 256) { run<0, true>(256); run<4, true>(256); run<8, true>(256); run<16, true>(256); run<32, true>(256); run<16, false>(256); run<32, false>(256); }
        else { run<0, true>(512); run<4, true>(512); run<8, true>(512); run<16, true>(512); run<32, true>(512); run<16, false>(512); run<32, false>(512); }
    }
    return 0;
}

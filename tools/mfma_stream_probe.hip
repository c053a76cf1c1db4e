// Diagnostics (GPU box): how much of the bf16 matrix pipe ONE wavefront per SIMD sustains when its MFMA stream is
// interleaved with LDS / global loads the way the cnn-trad conv2 loop is (in-order issue within a wavefront).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_stream_probe tools/mfma_stream_probe.hip
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

// MODE bit 0: one ds_read_b128 behind every second MFMA; bit 1: one global_load_dwordx4 behind every sixth MFMA;
// bit 2: the loaded values feed the following MFMAs (waits on them), otherwise they are only kept alive
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const uintx4* g, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uintx4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = g[i];
    __syncthreads();
    floatx16 acc0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, acc1 = acc0, acc2 = acc0;
    uintx4 a = g[lane], b0 = g[64 + lane], b1 = g[128 + lane], b2 = g[192 + lane];
    const uintx4* gp = g + wave * 64 + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uintx4 n0 = b0, n1 = b1, n2 = b2, na = a;
#define MF(acc, bb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), acc, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
        MF(acc0, b0)
        if (MODE & 1) { n0 = lds[(it * 3 & 15) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
        MF(acc1, b1)
        MF(acc2, b2)
        if (MODE & 1) { n1 = lds[((it * 3 + 1) & 15) * 64 + lane + 1024]; __builtin_amdgcn_sched_barrier(0); }
        MF(acc0, b1)
        MF(acc1, b2)
        if (MODE & 1) { n2 = lds[((it * 3 + 2) & 15) * 64 + lane + 2048]; __builtin_amdgcn_sched_barrier(0); }
        MF(acc2, b0)
        if (MODE & 2) { na = gp[(it & 31) * 512]; __builtin_amdgcn_sched_barrier(0); }
        if (MODE & 4) { b0 = n0; b1 = n1; b2 = n2; a = na; }
        else { asm volatile("" :: "v"(n0), "v"(n1), "v"(n2), "v"(na)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int MODE, int WAVES>
void run(const char* name) {
    const int grid = 256, iters = 512;
    uintx4* g; float* out; unsigned long long* cyc;
    hipMalloc(&g, sizeof(uintx4) * (32 * 512 + 4096));
    hipMemset(g, 0, sizeof(uintx4) * (32 * 512 + 4096));
    hipMalloc(&out, sizeof(float) * grid * WAVES * 64);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid * WAVES);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, g, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * WAVES);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= h.size();
    printf("%-62s %d wave(s)/SIMD: %.1f cycles per MFMA per wavefront, %.1f per SIMD\n", name, WAVES / 4, m / (iters * 6.0), m / (iters * 6.0) / (WAVES / 4));
    hipFree(g); hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 4>("MFMA only");
    run<1, 4>("+ 1 ds_read_b128 per 2 MFMAs (unused)");
    run<3, 4>("+ LDS reads + 1 global_load_dwordx4 per 6 MFMAs (unused)");
    run<5, 4>("+ LDS reads feeding the next iteration's MFMAs");
    run<7, 4>("+ LDS and global loads feeding the next iteration");
    run<0, 8>("MFMA only");
    run<7, 8>("+ LDS and global loads feeding the next iteration");
    return 0;
}

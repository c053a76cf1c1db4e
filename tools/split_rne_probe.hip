// Diagnostics (GPU box): is the gfx950 pair v_cvt_pk_bf16_f32 / v_dot2_f32_bf16 an exact way to split f32 into three
// bf16 pieces (hi = RNE(y), r1 = y - hi via dot2 with (-1, 0), mid = RNE(r1), ...), and at what rate does it issue?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/split_rne_probe tools/split_rne_probe.hip && tools/bin/split_rne_probe
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    uint32_t r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float minus_lo(uint32_t pk, float c) {
    float r;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(pk), "s"(0x0000BF80u), "v"(c));
    return r;
}
__device__ __forceinline__ float minus_hi(uint32_t pk, float c) {
    float r;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(pk), "s"(0xBF800000u), "v"(c));
    return r;
}

__global__ void split_kernel(const float* x, int n, uint32_t* pieces, float* rem) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    uint32_t hi = cvt_pk(a, b);
    // gfx940+ hazard: a DOT result read by a different VALU instruction needs 3 wait states (inline asm hides it)
    float a1 = minus_lo(hi, a), b1 = minus_hi(hi, b);
    asm volatile("s_nop 2" ::: "memory");
    uint32_t mid = cvt_pk(a1, b1);
    float a2 = minus_lo(mid, a1), b2 = minus_hi(mid, b1);
    asm volatile("s_nop 2" ::: "memory");
    uint32_t lo = cvt_pk(a2, b2);
    pieces[3 * i] = hi; pieces[3 * i + 1] = mid; pieces[3 * i + 2] = lo;
    rem[4 * i] = a1; rem[4 * i + 1] = b1; rem[4 * i + 2] = a2; rem[4 * i + 3] = b2;
}

template <int KIND>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    uint32_t pk = 0x3f803f80u + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[j]) : "v"(pk), "s"(0x0000BF80u));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(pk), "s"(0x3f800000u));
            if (KIND == 2) { uint32_t r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(v[j])); v[j] = __uint_as_float(r); }
            if (KIND == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[j]) : "s"(0xffff0000u));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float bf16_to_float(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        int e = 127 - 40 + rand() % 80;  // exponents 2^-40 .. 2^39
        u = (u & 0x807fffffu) | ((uint32_t)e << 23);
        memcpy(&x[i], &u, 4);
        if (i % 97 == 0) x[i] = 0.f;
        if (i % 101 == 0) x[i] = ldexpf(1.f, -20) * (rand() % 7);
    }
    float* dx; uint32_t* dp; float* dr;
    hipMalloc(&dx, n * 4); hipMalloc(&dp, 3 * (n / 2) * 4); hipMalloc(&dr, 2 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    split_kernel<<<n / 2 / 256, 256>>>(dx, n, dp, dr);
    std::vector<uint32_t> p(3 * (n / 2)); std::vector<float> r(2 * n);
    hipMemcpy(p.data(), dp, p.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, r.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i)
        printf("x = %.9g, %.9g  hi %08x mid %08x lo %08x  r1 %.9g %.9g r2 %.9g %.9g\n", x[2 * i], x[2 * i + 1], p[3 * i], p[3 * i + 1],
               p[3 * i + 2], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    long bad_rem = 0, bad_sum = 0; double worst = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int h = 0; h < 2; ++h) {
            double y = x[2 * i + h];
            double hi = bf16_to_float((uint16_t)(p[3 * i] >> (16 * h))), mid = bf16_to_float((uint16_t)(p[3 * i + 1] >> (16 * h))),
                   lo = bf16_to_float((uint16_t)(p[3 * i + 2] >> (16 * h)));
            double r1 = r[4 * i + h], r2 = r[4 * i + 2 + h];
            if (r1 != y - hi || r2 != r1 - mid) {
                if (bad_rem < 6) printf("  inexact: y %.9g hi %.9g r1 %.9g (want %.9g) mid %.9g r2 %.9g (want %.9g)\n", y, hi, r1, y - hi, mid, r2, r1 - mid);
                ++bad_rem;
            }
            double err = fabs(y - (hi + mid + lo));
            if (y != 0 && err / fabs(y) > worst) worst = err / fabs(y);
            if (err > ldexp(fabs(y), -24)) ++bad_sum;
        }
    printf("split of %d values: inexact remainders %ld, |y-(hi+mid+lo)| > 2^-24|y|: %ld, worst relative %.3g (2^-24 = %.3g)\n", n, bad_rem,
           bad_sum, worst, ldexp(1.0, -24));
    float* dout; unsigned long long* dc; hipMalloc(&dout, 64 * 4); hipMalloc(&dc, 8);
    const char* names[4] = {"v_dot2_f32_bf16", "v_fma_f32", "v_cvt_pk_bf16_f32", "v_and_b32"};
    for (int k = 0; k < 4; ++k) {
        unsigned long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (k == 0) rate_kernel<0><<<1, 64>>>(dout, dc, 4096);
            if (k == 1) rate_kernel<1><<<1, 64>>>(dout, dc, 4096);
            if (k == 2) rate_kernel<2><<<1, 64>>>(dout, dc, 4096);
            if (k == 3) rate_kernel<3><<<1, 64>>>(dout, dc, 4096);
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-20s %.2f cycles per instruction (one wavefront, 8 independent chains)\n", names[k], (double)c / (4096.0 * 8));
    }
    return 0;
}

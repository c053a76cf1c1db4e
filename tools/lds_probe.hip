// Micro-benchmark (diagnostics): LDS bank-conflict cycles of candidate access patterns, one kernel per pattern so
// that rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS attributes the counters.
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d out -- ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

// PATTERN -> byte address of this lane's access number c (0..7)
template <int PATTERN>
__device__ __forceinline__ int addr_of(int lane, int c) {
    const int k1 = lane >> 3, q = lane & 7;
    switch (PATTERN) {
        case 0: return 8 * (c * 72 + lane);                                  // contiguous rows (exchange 1 write)
        case 1: return 8 * (k1 * 72 + 8 * c + (q ^ (2 * (k1 & 3))));          // exchange 2 write, swizzled, row 72
        case 2: return 8 * (k1 * 66 + 8 * c + q);                             // exchange 2 write, row 66, no swizzle
        case 3: return 8 * (k1 * 72 + 8 * c + q);                             // row 72, no swizzle
        case 4: return 8 * (lane * 9 + c);                                    // lane-major, lane stride 9 complex
        case 5: return 8 * (k1 * 72 + 8 * c + q) + 64 * (k1 & 1);             // row 72 + 64-byte skew on odd rows
        case 6: return 8 * (k1 * 68 + 8 * c + q);                             // row 68
        case 7: return 8 * (k1 * 80 + 8 * c + q);                             // row 80
        default: return 0;
    }
}

template <int PATTERN, int KIND>  // KIND 0: ds_write_b64, 1: ds_read_b64
__global__ void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
    const int lane = threadIdx.x & 63;
    floatx2 v = {(float)lane, 1.0f};
    floatx2 acc = {0.f, 0.f};
    for (int i = threadIdx.x; i < 8 * 1024; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned base = lds_addr(lds) + (threadIdx.x >> 6) * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned a = base + addr_of<PATTERN>(lane, c);
            if (KIND == 0)
                asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
            else {
                floatx2 r;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
                acc += r;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + lds[threadIdx.x];
}

template <int PATTERN, int KIND>
void run() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256);
    hipLaunchKernelGGL((probe<PATTERN, KIND>), dim3(256), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipFree(out);
}

int main() {
    run<0, 0>(); run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>(); run<5, 0>(); run<6, 0>(); run<7, 0>();
    run<0, 1>(); run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>();
    printf("done\n");
    return 0;
}

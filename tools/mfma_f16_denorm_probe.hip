// Probe (GPU box): does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs, or flush them to zero?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_denorm_probe.hip -o tools/bin/mfma_f16_denorm_probe
// A[i][k] = a for k == 0 else 0; B[k][j] = b for k == 0 else 0  ->  D[i][j] = a * b.  Cases: a subnormal (2^-20) x b = 2^10,
// a normal x b subnormal, both normal (control), a = smallest subnormal 2^-24.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__global__ void probe(float a, float b, float* out) {
    halfx8 A = {0, 0, 0, 0, 0, 0, 0, 0}, B = A;
    if (threadIdx.x < 32) {  // k = 8 * (lane >> 5) + j: lanes 0..31 hold k = 0..7
        A[0] = (_Float16)a;
        B[0] = (_Float16)b;
    }
    floatx16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {1024.f, ldexpf(1.f, -20)}, {ldexpf(1.f, -10), 1024.f}, {ldexpf(1.f, -24), 4096.f},
                              {ldexpf(3.f, -16), 1.f}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, cs[0], cs[1], d);
        float h = -1.f;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.9g  b = %.9g  expected a*b = %.9g  mfma = %.9g  %s\n", cs[0], cs[1], cs[0] * cs[1], h, h == cs[0] * cs[1] ? "exact" : (h == 0.f ? "FLUSHED" : "other"));
    }
    return 0;
}

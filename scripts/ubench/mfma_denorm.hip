// Probe: does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs (A side, B side) and fp32 subnormal accumulators?
// One wave.  Prints the accumulator for A = 2^-20 (subnormal fp16) x B = 1, A = 1 x B = 2^-20, and both normal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
    h8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)a; bv[j] = (_Float16)b; }
    f16v acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; if (hipMalloc(&d, 4) != hipSuccess) return 1;
    const float cases[4][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {0.5f, 2.0f}, {9.5367431640625e-07f, 16384.0f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float h = -1; if (hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        printf("a=%g b=%g -> acc=%g (expected %g)\n", c[0], c[1], h, 16.0 * c[0] * c[1]);
    }
    return 0;
}

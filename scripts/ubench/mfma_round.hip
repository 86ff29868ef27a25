// Probe: how does v_mfma_f32_32x32x16_f16 round D = C + sum_16 a b ?  The products of fp16 numbers are exact in fp32, the
// sum of 16 of them plus C is evaluated on the host in long double (exact for these magnitudes) and D is compared with it
// in units of 2^-24 x (|C| + sum |a b|), the magnitude the adder works at (relative to |D| itself the error is unbounded
// under cancellation, as for any summation): <= 1 everywhere means no worse than one rounding to nearest at that magnitude,
// <= 2 a truncation, more an internal format that loses bits.  The FIR error budgets (scripts/fir_error_budget.py) charge `c` x 2^-24 x |partial sum| per MFMA
// of a chain; this prints the c to use.  One wave per case.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// A[32][16], B[16][32] row-major fp16, C/D[32][32] fp32
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    h8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = A[n * 16 + 8 * h + j]; bv[j] = B[(8 * h + j) * 32 + n]; }
    f16v acc;
    for (int e = 0; e < 16; ++e) acc[e] = C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + n];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const float v = acc[e];
        D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + n] = v;
    }
}
int main() {
    _Float16 *dA, *dB; float *dC, *dD;
    if (hipMalloc(&dA, 32 * 16 * 2) != hipSuccess || hipMalloc(&dB, 16 * 32 * 2) != hipSuccess ||
        hipMalloc(&dC, 32 * 32 * 4) != hipSuccess || hipMalloc(&dD, 32 * 32 * 4) != hipSuccess) return 1;
    std::mt19937 rng(12345);
    // cases: spread of product magnitudes (exponent range of a and b) and size of C relative to the sum
    const struct { int spread; float cscale; const char* what; } cases[] = {
        {0, 0.0f, "equal magnitudes, C = 0"}, {0, 1.0f, "equal magnitudes, |C| ~ |sum|"}, {0, 64.0f, "equal magnitudes, |C| ~ 64 |sum|"},
        {6, 1.0f, "products over 12 binades, |C| ~ |sum|"}, {10, 1.0f, "products over 20 binades, |C| ~ |sum|"},
        {10, 1024.0f, "products over 20 binades, |C| ~ 1024 |sum|"}, {3, 0.01f, "products over 6 binades, |C| ~ 0.01 |sum|"}};
    for (const auto& cs : cases) {
        double worst = 0, sum_ulp = 0; long cnt = 0, inexact = 0;
        for (int rep = 0; rep < 200; ++rep) {
            std::vector<_Float16> A(32 * 16), B(16 * 32);
            std::vector<float> C(32 * 32), D(32 * 32);
            std::uniform_real_distribution<float> u(-1.0f, 1.0f);
            std::uniform_int_distribution<int> ex(-cs.spread, cs.spread);
            for (auto& v : A) v = (_Float16)(u(rng) * std::ldexp(1.0f, ex(rng)));
            for (auto& v : B) v = (_Float16)(u(rng) * std::ldexp(1.0f, ex(rng)));
            for (auto& v : C) v = u(rng) * cs.cscale * 2.0f;
            hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    long double s = C[r * 32 + c], mag = fabsl((long double)C[r * 32 + c]);
                    for (int kk = 0; kk < 16; ++kk) {
                        const long double pr = (long double)(float)A[r * 16 + kk] * (long double)(float)B[kk * 32 + c];
                        s += pr;
                        mag += fabsl(pr);
                    }
                    const float d = D[r * 32 + c];
                    const long double ulp = mag * std::ldexp(1.0L, -24);
                    const double err = (double)(fabsl((long double)d - s) / ulp);
                    if (err > worst) worst = err;
                    sum_ulp += err; ++cnt;
                    if ((long double)(float)s != s) ++inexact;
                }
        }
        printf("%-48s max |D - exact| = %.3f x 2^-24 (|C| + sum |a b|), mean %.3f, %ld of %ld sums not representable in fp32\n", cs.what, worst,
               sum_ulp / cnt, inexact, cnt);
    }
    return 0;
}

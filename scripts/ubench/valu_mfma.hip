// Microbenchmark: do VALU work and MFMA work of the waves on one SIMD overlap?  Each wave loops over "K-steps" of NV
// packed-FMA instructions (independent chains) followed by two MFMA 32x32x16 f16 that accumulate into the same registers
// (the FIR kernels' pattern; the MFMA operands do not depend on the VALU results here, so only the issue order couples
// them).  Compared at 1, 2 and 3 waves per SIMD: VALU only, MFMA only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int NV, bool MF, bool DEP, int OP = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2f a[10];
    for (int i = 0; i < 10; ++i) a[i] = (v2f){1.0f + i + threadIdx.x * 1e-3f, 0.5f};
    const v2f m = {0.999f, 1e-3f};
    v8h av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(0.01f * j + threadIdx.x * 1e-4f); bv[j] = (_Float16)(0.02f * j); }
    v16f acc = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (OP == 0) a[i % 10] = __builtin_elementwise_fma(a[i % 10], m, a[i % 10]);          // v_pk_fma_f32
            else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i % 10].x) : "v"(m.x));   // plain fp32
            else if (OP == 2) { unsigned r; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i % 10].x), "v"(a[i % 10].y)); a[i % 10].x = __uint_as_float(r | 0x3c000000u); }
            else asm volatile("v_fma_mix_f32 %0, %0, %1, -%1 op_sel_hi:[0,0,1]" : "+v"(a[i % 10].x) : "v"(m.x));
        }
        if (MF) {
            if (DEP) {   // the A operand comes out of the VALU work of this step, as in the FIR kernels
                av[0] = (_Float16)a[0].x; av[1] = (_Float16)a[1].x;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, acc, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 10; ++i) s += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool MF, bool DEP, int OP = 0>
float run(float* out, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, MF, DEP, OP>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int wps : {1, 2, 3}) {
        const int blocks = 256 * wps;
        const float v = run<20, false, false>(out, blocks), mf = run<0, true, false>(out, blocks),
                    both = run<20, true, false>(out, blocks), dep = run<20, true, true>(out, blocks);
        printf("%d wave(s)/SIMD: 20 pk_fma per step %.3f ms, 2 MFMA per step %.3f ms, both %.3f ms, both with A from the VALU %.3f ms\n",
               wps, v, mf, both, dep);
        fflush(stdout);
        const char* names[] = {"v_pk_fma_f32", "v_fma_f32", "v_cvt_pkrtz_f16_f32 (+v_or)", "v_fma_mix_f32"};
        const float vo[4] = {v, run<20, false, false, 1>(out, blocks), run<20, false, false, 2>(out, blocks), run<20, false, false, 3>(out, blocks)};
        const float bo[4] = {both, run<20, true, false, 1>(out, blocks), run<20, true, false, 2>(out, blocks), run<20, true, false, 3>(out, blocks)};
        for (int o = 0; o < 4; ++o) printf("    20 x %-28s alone %.3f ms, with the 2 MFMA %.3f ms (sum would be %.3f)\n", names[o], vo[o], bo[o], vo[o] + mf);
        fflush(stdout);
    }
    return 0;
}

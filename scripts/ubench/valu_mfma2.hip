// Microbenchmark: which vector instructions run under an MFMA of the SAME wave (one wave per SIMD, the wide FIR's shape)?
// Per step: one v_mfma_f32_32x32x16_f16 followed by NV independent vector instructions of one kind.  Times: VALU only,
// MFMA only, both -- "both" near max(...) means the kind overlaps with the matrix pipe, near the sum means it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int OP>
__device__ __forceinline__ void vop(unsigned& d, unsigned a, unsigned b) {
    if (OP == 0) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 1) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 6) asm volatile("v_fma_mixhi_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 7) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 8) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 9) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 10) asm volatile("v_mul_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 11) asm volatile("v_fma_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 12) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(unsigned long long*)&d) : "v"((unsigned long long)a), "v"((unsigned long long)b));
    else if (OP == 13) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (OP == 14) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else if (OP == 15) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(d) : "v"(a));
}

template <int OP, int NV, bool VA, bool MF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = 0x3c003c00u + i + threadIdx.x;
    const unsigned a = 0x3c013c01u + threadIdx.x, b = 0x38003800u;
    v8h av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(0.01f * j + threadIdx.x * 1e-4f); bv[j] = (_Float16)(0.02f * j); }
    v16f acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MF) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[m & 1], 0, 0, 0);
            if (VA) {
#pragma unroll
                for (int i = 0; i < NV; ++i) vop<OP>(r[(m * NV + i) & 15], a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += __uint_as_float(r[i]) + acc[0][i] + acc[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int NV, bool VA, bool MF>
float run(float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, NV, VA, MF>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = -1; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int OP>
void test(const char* name, float* d, float mf) {
    const int iters = 20000;
    constexpr int NV = 6;
    const float va = run<OP, NV, true, false>(d, iters), both = run<OP, NV, true, true>(d, iters);
    printf("%-22s x%d per MFMA: alone %.3f ms (%.2f cyc/instr at 2.4 GHz)  MFMA alone %.3f  both %.3f  (sum %.3f, max %.3f)  overlap %.0f %%\n", name, NV, va,
           va * 1e-3 * 2.4e9 / (iters * 4.0 * NV), mf, both, va + mf, va > mf ? va : mf, 100.0 * (va + mf - both) / (va < mf ? va : mf));
    fflush(stdout);
}

int main() {
    float* d; if (hipMalloc(&d, 256 * 256 * 4) != hipSuccess) return 1;
    const float mf = run<0, 6, false, true>(d, 20000);
    printf("MFMA alone: %.3f ms = %.1f cycles per MFMA at 2.4 GHz\n", mf, mf * 1e-3 * 2.4e9 / (20000 * 4.0));
    test<0>("v_pk_mul_f16", d, mf); test<1>("v_pk_fma_f16", d, mf); test<2>("v_mul_f32", d, mf); test<3>("v_fma_f32", d, mf);
    test<4>("v_fma_mix_f32", d, mf); test<5>("v_fma_mixlo_f16", d, mf); test<6>("v_fma_mixhi_f16", d, mf); test<7>("v_cvt_pk_f16_f32", d, mf);
    test<8>("v_cvt_pkrtz_f16_f32", d, mf); test<9>("v_perm_b32", d, mf); test<10>("v_mul_f16", d, mf); test<11>("v_fma_f16", d, mf);
    test<12>("v_pk_mul_f32", d, mf); test<13>("v_xor_b32", d, mf); test<14>("v_and_or_b32", d, mf); test<15>("v_cvt_f16_f32", d, mf);
    return 0;
}

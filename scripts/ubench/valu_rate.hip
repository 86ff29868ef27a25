// Microbenchmark: issue cost of the VALU instructions of the FIR kernels' weight generation, 2 waves per SIMD, long
// independent chains (8 accumulators per lane), so that the number is the instruction's issue/execute rate and not a latency.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = seed * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) {          // v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 1) {   // v_pk_mul_f32 (two registers)
                v2f x = {a[i], b[i]};
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(x));
                a[i] = x.x; b[i] = x.y;
            } else if (OP == 2) {   // v_cvt_pkrtz_f16_f32
                unsigned r;
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i]));
                a[i] = __uint_as_float(r);
            } else if (OP == 3) {   // v_fma_mix_f32 (f32 * f32 - f16)
                asm volatile("v_fma_mix_f32 %0, %0, %1, -%1 op_sel_hi:[0,0,1]" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 4) {   // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 5) {   // v_perm_b32
                asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (OP == 6) {   // v_pk_fma_f16
                asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;       // 2 blocks of 4 waves per CU = 2 waves per SIMD
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 2 waves x iters x 8 instructions
    printf("%-22s %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, best, best * 1e6 / (2.0 * iters * 8));
}

int main() {
    float* out; (void)hipMalloc(&out, 512 * 256 * 4);
    run<0>("v_mul_f32", out); run<1>("v_pk_mul_f32", out); run<2>("v_cvt_pkrtz_f16_f32", out); run<3>("v_fma_mix_f32", out);
    run<4>("v_fma_f32", out); run<5>("v_perm_b32", out); run<6>("v_pk_fma_f16", out);
    return 0;
}

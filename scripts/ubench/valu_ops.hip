// Microbenchmark: issue rate of the packed-fp32 forms the FFT uses (2 waves/SIMD, 4 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(v2f* out, int iters, v2f m) {                  \
        v2f a[4];                                                                              \
        for (int i = 0; i < 4; ++i) a[i] = (v2f){(float)i + threadIdx.x * 1e-3f, 1.f};        \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                   \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(m)); \
            }                                                                                  \
        }                                                                                      \
        v2f s = {0, 0};                                                                        \
        for (int i = 0; i < 4; ++i) s += a[i];                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                               \
    }
KERNEL(k_fma, "v_pk_fma_f32 %0, %0, %1, %0")
KERNEL(k_fma_sel, "v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]")
KERNEL(k_add, "v_pk_add_f32 %0, %0, %1")
KERNEL(k_add_sel, "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
KERNEL(k_mul, "v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]")

template <typename K>
void run(const char* name, K kern) {
    const int blocks = 256 * 2, iters = 20000;
    v2f* d; (void)hipMalloc(&d, sizeof(v2f) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, (v2f){0.999f, 1e-3f});
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-12s %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, best, best * 1e6 / (64.0 * iters * 2));
    (void)hipFree(d);
}
int main() {
    run("pk_fma", k_fma); run("pk_fma_sel", k_fma_sel); run("pk_add", k_add); run("pk_add_sel", k_add_sel);
    run("pk_mul", k_mul);
    return 0;
}

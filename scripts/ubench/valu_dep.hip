// Microbenchmark: v_pk_fma_f32 issue rate vs instruction-level parallelism (independent chains per wave) and waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int ILP>
__global__ __launch_bounds__(256) void k(v2f* out, int iters, v2f m) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)i + threadIdx.x * 1e-3f, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / ILP; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
        }
    }
    v2f s = {0, 0};
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ILP>
void run(int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    v2f* d; (void)hipMalloc(&d, sizeof(v2f) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, (v2f){0.999f, 1e-9f});
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double insts = 64.0 * iters;                       // per wave
    // each wave sits on one SIMD with (blocks_per_cu) waves; time per instruction per SIMD:
    printf("ILP %d  waves/SIMD %d: %.3f ms  %.2f ns per wave-instruction slot (%.1f TFLOP/s)\n", ILP, blocks_per_cu, best,
           best * 1e6 / (insts * blocks_per_cu), 4.0 * 64 * insts * blocks * 4 / best / 1e9);
    (void)hipFree(d);
}
int main() {
    for (int b : {1, 2, 4}) { run<1>(b); run<2>(b); run<3>(b); run<4>(b); run<8>(b); }
    return 0;
}

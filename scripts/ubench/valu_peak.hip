// Microbenchmark: what fp32 FMA rate does MI355X actually hold under a dense packed-FMA stream?
// Build: hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak ; run: ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(v2f* out, int iters, unsigned long long tap) {
    v2f a[8];
    v2f x = {1.0f + threadIdx.x * 1e-6f, 0.5f};
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "s"(tap));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x));
                if (MODE == 2) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(x.x), "s"((unsigned)tap));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(x.y), "s"((unsigned)tap));
                }
            }
        }
    }
    v2f s = {0, 0};
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    v2f* d;
    hipMalloc(&d, sizeof(v2f) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float tap = 1e-9f; unsigned long long tp; unsigned t; memcpy(&t, &tap, 4); tp = ((unsigned long long)t << 32) | t;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, tp);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 4.0 * 128.0 * iters * 256.0 * blocks;  // 128 packed FMAs (=4 flop) per iteration per thread
        printf("%s blocks/CU=%d rep%d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, rep, ms, flops / ms / 1e9);
    }
    hipFree(d);
}
#include <cstring>
int main() {
    for (int b : {1, 2, 4, 8}) {
        run<0>("pk_fma sgpr-tap", b);
        run<1>("pk_fma vgpr   ", b);
        run<2>("2x v_fma sgpr ", b);
    }
    return 0;
}

// Microbenchmark: packed-FMA rate and held clock with RANDOM operands (data toggling raises power -> DVFS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k(v2f* out, const v2f* in, int iters, const unsigned long long* taps) {
    v2f a[8], x[8];
    unsigned long long tp[8];
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){0.f, 0.f}; x[i] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + i]; tp[i] = taps[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x[(i + u) & 7]), "s"(tp[u & 7]));
        }
    }
    v2f s = {0, 0};
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;  // 0 = constant data, 1 = random data
    for (int bpc : {4, 8}) {
        const int blocks = 256 * bpc, iters = 20000;
        std::vector<float> h((size_t)blocks * 256 * 16);
        for (auto& v : h) v = mode ? (float)rand() / RAND_MAX * 14.f - 7.f : 1.0f;
        float ht[16];
        for (auto& v : ht) v = mode ? ((float)rand() / RAND_MAX - 0.5f) * 1e-3f : 0.f;   // tiny taps keep values finite
        v2f *d, *din; unsigned long long* dt;
        hipMalloc(&d, sizeof(v2f) * blocks * 256); hipMalloc(&din, h.size() * 4); hipMalloc(&dt, 64);
        hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, ht, 64, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, din, iters, dt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d blocks/CU=%d rep%d: %.3f ms %.1f TFLOP/s\n", mode, bpc, rep, ms, 4.0 * 128 * iters * 256.0 * blocks / ms / 1e9);
        }
    }
    return 0;
}

// Microbenchmark: what streaming-store rate does MI355X hold?  dwordx2 vs dwordx4 per lane, plain vs nontemporal,
// and the FIR's own pattern (7 KB chunks, 28 rows of 256 B per half-wave).
// Build: hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int W, bool NT>
__global__ __launch_bounds__(256) void fill(float* out, size_t n_elems_per_thread_block) {
    // each block writes a contiguous chunk; each wave instruction writes 64 * W * 4 contiguous bytes
    const size_t chunk = n_elems_per_thread_block;  // floats per block
    float* p = out + (size_t)blockIdx.x * chunk + threadIdx.x * W;
    const float v = (float)threadIdx.x;
    for (size_t i = 0; i < chunk; i += 256 * W) {
        if (W == 2) { v2f x = {v, v}; if (NT) __builtin_nontemporal_store(x, (v2f*)(p + i)); else *(v2f*)(p + i) = x; }
        if (W == 4) { v4f x = {v, v, v, v}; if (NT) __builtin_nontemporal_store(x, (v4f*)(p + i)); else *(v4f*)(p + i) = x; }
    }
}

// FIR-like: half-wave writes 28 rows of 256 B (7168 B chunk), tasks scattered over (channel, field) like the kernel
__global__ __launch_bounds__(512) void fir_like(float* out, int chunks_per_field, int channels, int fields, int cpt) {
    const int t = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const long task = (long)blockIdx.x * 16 + hw;
    const int b = task % chunks_per_field;
    const long g = task / chunks_per_field;
    const int groups = channels / cpt;
    const int cg = g % groups, f = g / groups;
    if (f >= fields) return;
    const size_t field_floats = (size_t)chunks_per_field * 1792;
    for (int c = cg * cpt; c < (cg + 1) * cpt; ++c) {
        float* dst = out + ((size_t)c * fields + f) * field_floats + (size_t)b * 1792 + 2 * t;
#pragma unroll
        for (int n = 0; n < 28; ++n) { v2f x = {(float)n, (float)c}; *(v2f*)(dst + 64 * n) = x; }
    }
}

template <typename F>
void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s rep%d: %.3f ms  %.2f TB/s\n", name, rep, ms, bytes / ms / 1e9);
    }
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB
    float* d; hipMalloc(&d, bytes);
    const int blocks = 8192;
    const size_t chunk = bytes / 4 / blocks;
    timeit("dwordx2 plain", bytes, [&] { hipLaunchKernelGGL((fill<2, false>), dim3(blocks), dim3(256), 0, 0, d, chunk); });
    timeit("dwordx2 nt", bytes, [&] { hipLaunchKernelGGL((fill<2, true>), dim3(blocks), dim3(256), 0, 0, d, chunk); });
    timeit("dwordx4 plain", bytes, [&] { hipLaunchKernelGGL((fill<4, false>), dim3(blocks), dim3(256), 0, 0, d, chunk); });
    timeit("dwordx4 nt", bytes, [&] { hipLaunchKernelGGL((fill<4, true>), dim3(blocks), dim3(256), 0, 0, d, chunk); });
    const int cpf = 290, C = 64, F = 64, cpt = 8;
    const double fb = (double)cpf * 7168 * C * F;
    const long tasks = (long)cpf * (C / cpt) * F;
    timeit("fir-like 7KB chunks x2", fb, [&] { hipLaunchKernelGGL(fir_like, dim3((tasks + 15) / 16), dim3(512), 0, 0, d, cpf, C, F, cpt); });
    hipMemset(d, 0, bytes);
    timeit("hipMemset 8GiB", bytes, [&] { hipMemsetAsync(d, 0, bytes, 0); });
    return 0;
}

// Microbenchmark: does an 8.6 GB store stream overlap with arithmetic in the same waves?  Each wave alternates a burst
// of packed-FMA work (registers only) with 8 coalesced 1 KB stores, like the FIR kernels.  Compared: stores only, compute
// only (stores into a 1 KB dummy), both.  If "both" ~ max(stores, compute) the two overlap; if ~ sum they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool REAL_STORES>
__global__ __launch_bounds__(256) void k(v4f* __restrict__ dst, v4f* __restrict__ dummy, long nchunks, int fmas) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)i + lane * 1e-3f, 1.f};
    const v2f m = {0.999f, 1e-3f};
    for (long c = wave; c < nchunks; c += nwaves) {
        for (int it = 0; it < fmas; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, a[i]);
        v4f v = {a[0].x + a[1].x, a[2].x + a[3].x, a[4].x + a[5].x, a[6].x + a[7].x};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v4f* p = REAL_STORES ? &dst[(c * 8 + r) * 64 + lane] : &dummy[lane];
            *p = v + (float)r;
        }
    }
}

template <bool REAL>
float run(const char* name, v4f* dst, v4f* dummy, long nchunks, int fmas, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<REAL>), dim3(blocks), dim3(256), 0, 0, dst, dummy, nchunks, fmas);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms\n", name, best);
    return best;
}

int main() {
    const long bytes = 8L << 30, nchunks = bytes / 8192;
    v4f *dst, *dummy;
    (void)hipMalloc(&dst, bytes); (void)hipMalloc(&dummy, 4096);
    for (int blocks : {256 * 2, 256 * 3}) {       // 8 or 12 waves per CU, like the FIR kernels
        printf("-- %d waves per CU\n", blocks / 256 * 4);
        run<true>("stores only", dst, dummy, nchunks, 0, blocks);
        for (int fmas : {40, 80, 120}) {
            char n[64];
            snprintf(n, sizeof n, "compute only (%d x 8 pk_fma per 8 KB)", fmas);
            run<false>(n, dst, dummy, nchunks, fmas, blocks);
            snprintf(n, sizeof n, "compute + stores (%d x 8 pk_fma per 8 KB)", fmas);
            run<true>(n, dst, dummy, nchunks, fmas, blocks);
        }
    }
    return 0;
}

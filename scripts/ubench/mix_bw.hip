// Microbenchmark: what does MI355X sustain for the FIR's HBM traffic MIX -- an 8.5 GB write stream with a 1.07 GB (or
// larger) read stream beside it -- as opposed to a pure write stream (write_bw.hip)?  Every wave alternates: one
// 16-byte-per-lane load of fresh data, then R 16-byte-per-lane stores (R = 8: the 1:8 byte ratio of symbols : IQ).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int R, bool READ, bool NT = false>
__global__ __launch_bounds__(512) void k_mix(const v4f* __restrict__ src, v4f* __restrict__ dst, long nchunks) {
    // chunk = 64 lanes x 16 B x R stores = R KB written, 1 KB read
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long c = wave; c < nchunks; c += nwaves) {
        v4f v = {1.f, 2.f, 3.f, 4.f};
        if (READ) v = src[c * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (NT) __builtin_nontemporal_store(v + (float)r, &dst[(c * R + r) * 64 + lane]);
            else dst[(c * R + r) * 64 + lane] = v + (float)r;
        }
    }
}

template <int R, bool READ, bool NT = false>
void run(const char* name, const v4f* src, v4f* dst, long nchunks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<R, READ, NT>), dim3(256 * 4), dim3(512), 0, 0, src, dst, nchunks);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double wr = (double)nchunks * R * 1024, rd = READ ? (double)nchunks * 1024 : 0;
        printf("%-28s rep%d: %.3f ms  write %.2f TB/s  total %.2f TB/s\n", name, rep, ms, wr / ms / 1e9, (wr + rd) / ms / 1e9);
    }
}

int main() {
    const long wr_bytes = 8L << 30;
    v4f *src, *dst;
    (void)hipMalloc(&dst, wr_bytes);
    (void)hipMalloc(&src, wr_bytes / 2);
    (void)hipMemset(src, 1, wr_bytes / 2);
    run<8, false>("write only (8 KB chunks)", src, dst, wr_bytes / 8192);
    run<8, true>("1 KB read : 8 KB write", src, dst, wr_bytes / 8192);
    run<8, true, true>("1 KB read : 8 KB write, nt", src, dst, wr_bytes / 8192);
    run<8, false, true>("write only, nt", src, dst, wr_bytes / 8192);
    run<4, true>("1 KB read : 4 KB write", src, dst, wr_bytes / 4096);
    run<2, true>("1 KB read : 2 KB write", src, dst, wr_bytes / 2048);
    return 0;
}

// Microbenchmark: packed FMA stream with random operands AND an LDS read every N FMAs (the FIR's shape),
// optionally a 16-byte global store every M FMAs: which of them pulls the held clock down?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int FMA_PER_READ, int STORE, int COAL>
__global__ __launch_bounds__(256) void k(v2f* out, const v2f* in, int iters, const unsigned long long* taps, float4* sink) {
    __shared__ v2f s[2176];
    v2f a[8];
    unsigned long long tp[8];
    for (int i = threadIdx.x; i < 2176; i += 256) s[i] = in[(blockIdx.x * 2176 + i) % (1 << 20)];
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){0.f, 0.f}; tp[i] = taps[i]; }
    __syncthreads();
    const v2f* p = s + threadIdx.x;
    float4* dst = sink + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    float4* dstc = sink + (size_t)blockIdx.x * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);  // wave-contiguous 1 KB per instruction
    for (int it = 0; it < iters; ++it) {
        v2f x = p[0];
#pragma unroll
        for (int u = 0; u < 128 / FMA_PER_READ; ++u) {
            const v2f xn = p[((u + 1) * 17) % 1900];
#pragma unroll
            for (int i = 0; i < FMA_PER_READ; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i & 7]) : "v"(x), "s"(tp[(u + i) & 7]));
            x = xn;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STORE && (it % STORE) == 0) {  // 64 B per lane, like 8 complex outputs
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (COAL) dstc[r * 64] = make_float4(a[2 * r].x, a[2 * r].y, a[2 * r + 1].x, a[2 * r + 1].y);
                else dst[r] = make_float4(a[2 * r].x, a[2 * r].y, a[2 * r + 1].x, a[2 * r + 1].y);
            }
            dst += (size_t)gridDim.x * 256 * 4;
            dstc += (size_t)gridDim.x * 256 * 4;
        }
    }
    v2f t = {0, 0};
    for (int i = 0; i < 8; ++i) t += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int FPR, int STORE, int COAL = 0>
void run(const char* name) {
    const int blocks = 256 * 4, iters = 4000;
    std::vector<float> h((size_t)2 << 20);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 14.f - 7.f;
    float ht[16];
    for (auto& v : ht) v = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f;
    v2f *d, *din; unsigned long long* dt; float4* sink;
    hipMalloc(&d, sizeof(v2f) * blocks * 256); hipMalloc(&din, h.size() * 4); hipMalloc(&dt, 64);
    const size_t sink_elems = STORE ? (size_t)blocks * 256 * 4 * (iters / STORE + 1) : 16;
    hipMalloc(&sink, sink_elems * sizeof(float4));
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, ht, 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FPR, STORE, COAL>), dim3(blocks), dim3(256), 0, 0, d, din, iters, dt, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%s rep%d: %.3f ms %.1f TFLOP/s  store %.2f TB/s\n", name, rep, ms, 4.0 * 128 * iters * 256.0 * blocks / ms / 1e9,
               STORE ? 64.0 * 256 * blocks * (iters / STORE) / ms / 1e9 : 0.0);
    }
    hipFree(d); hipFree(din); hipFree(dt); hipFree(sink);
}
int main() {
    run<128, 0>("no LDS reads           ");
    run<8, 0>("1 ds_read_b64 / 8 fma  ");
    run<4, 0>("1 ds_read_b64 / 4 fma  ");
    run<16, 0>("1 ds_read_b64 / 16 fma ");
    run<8, 7>("read/8 + store 64B/896 ");
    run<8, 7, 1>("same, coalesced stores ");
    run<128, 7, 0>("no LDS, strided stores ");
    run<128, 7, 1>("no LDS, coalesced st.  ");
    return 0;
}

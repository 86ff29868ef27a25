// Microbenchmark: what bounds a pure store stream?  8 GiB written once by 4-wave blocks, every wave walks its own
// contiguous region.  Varied: bytes per lane and store instruction (16 / 8 / 4) and the number of blocks (256 CUs:
// 128 blocks leave half of them idle).  If the time follows the number of store INSTRUCTIONS the limit is per
// instruction, if it follows the number of active CUs it is per CU (L1 -> L2), if neither it is the memory behind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int W>   // bytes per lane per store
__global__ __launch_bounds__(256) void k(char* __restrict__ dst, long bytes_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    char* p = dst + wave * bytes_per_wave + lane * W;
    const v4f v = {(float)lane, 1.f, 2.f, 3.f};
    for (long o = 0; o < bytes_per_wave; o += 64 * W * 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            char* q = p + o + r * 64 * W;
            if (W == 16) *reinterpret_cast<v4f*>(q) = v;
            else if (W == 8) *reinterpret_cast<v2f*>(q) = (v2f){v.x, v.y};
            else *reinterpret_cast<float*>(q) = v.x;
        }
    }
}

template <int W>
void run(char* dst, long bytes, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long per = bytes / (blocks * 4L);      // bytes per wave: a multiple of 4 KB for every block count used
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(256), 0, 0, dst, per);
        hipError_t le = hipGetLastError();
        (void)hipEventRecord(e1); hipError_t se = hipEventSynchronize(e1);
        float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, e0, e1);
        if (le != hipSuccess || se != hipSuccess || ee != hipSuccess) { printf("error %s %s %s\n", hipGetErrorString(le), hipGetErrorString(se), hipGetErrorString(ee)); exit(1); }
        if (ms < best) best = ms;
    }
    printf("blocks=%4d  %2d B/lane  %.3f ms  %.2f TB/s  %.1f ns per store instruction and active CU\n", blocks, W, best,
           bytes / best / 1e9, best * 1e6 / (double(bytes) / (64 * W) / (blocks < 256 ? blocks : 256)));
    fflush(stdout);
}

int main() {
    const long bytes = 8L << 30;
    char* dst;
    if (hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int blocks : {64, 128, 256, 512, 1024, 2048}) {
        run<16>(dst, bytes, blocks);
        run<8>(dst, bytes, blocks);
        run<4>(dst, bytes, blocks);
    }
    return 0;
}

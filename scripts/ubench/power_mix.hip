// Microbenchmark: what do the two halves of the FIR kernel cost in POWER?  Three long-running loads, each about 4 s, to be
// sampled from outside with `amd-smi metric -p -c` (scripts/dev/power_ubench.sh):
//   mode 0  a pure non-temporal store stream (the FIR's IQ pattern size: 8.5 GB per pass)
//   mode 1  dense v_mfma_f32_32x32x16_f16 chains, 2 waves per SIMD, no memory traffic
//   mode 2  both in one kernel: every wave issues 27 MFMAs and then stores 4 KB, like a tile of the FIR
// Build: hipcc --offload-arch=gfx950 -O3 power_mix.hip -o bin/power_mix ; run: bin/power_mix <mode> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(char* __restrict__ dst, long tiles_per_wave, float seed) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed * 0.013f + lane * 0.0371f + j * 0.113f - 1.3f); b[j] = (_Float16)(seed * 0.5f - lane * 0.0217f + j * 0.0931f); }
    f16v acc = {};
    char* p = dst + wave * tiles_per_wave * 4096 + lane * 16;
    for (long t = 0; t < tiles_per_wave; ++t) {
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 27; ++i) acc = (i & 1) ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);   // (operands alternate: data toggles)
        }
        if (MODE != 1) {
            const v4f v = {acc[0], acc[1], acc[2], acc[3] + (float)t};
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p + t * 4096 + r * 1024));
        }
    }
    if (MODE == 1 && acc[5] == 12345.678f) *reinterpret_cast<float*>(dst) = acc[7];   // keep the chain alive
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    const long waves = 2048, tiles_per_wave = 1017;                  // 2048 x 1017 x 4 KB = 8.5 GB per pass
    const long bytes = waves * tiles_per_wave * 4096;
    char* dst;
    if (hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0; int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, dst, tiles_per_wave, 1.0f);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, dst, tiles_per_wave, 1.0f);
            else hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, dst, tiles_per_wave, 1.0f);
        }
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms_sum += ms; n += 20;
    }
    const double ms = ms_sum / n;
    printf("mode %d: %.3f ms per pass", mode, ms);
    if (mode != 1) printf(", %.2f TB/s", bytes / ms / 1e9);
    if (mode != 0) printf(", %.0f TFLOP/s fp16 (27 MFMAs per tile)", 27.0 * 2 * 32 * 32 * 16 * waves * tiles_per_wave / ms / 1e9);
    printf("\n");
    hipFree(dst);
    return 0;
}

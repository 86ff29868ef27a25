// Microbenchmark: the fused FIR's output in its present shape (per tile of 32 samples and 16 channel rows: 4 dwordx4 stores,
// a lane = 16 bytes, an instruction = 4 rows x 256 B, out of an LDS image) against the shape the TRANSPOSED product would
// have (operands swapped: a lane of the accumulator tile is an OUTPUT, register pairs are (re, im) of one channel, so the
// tile leaves as 8 dwordx2 stores straight from the accumulators, a lane = 8 bytes, an instruction = 2 rows x 256 B, no LDS
// image), both with the tile's 21 MFMAs between the bursts and non-temporal.  [64 channels][64 fields][260416] complex64,
// wave = (field, 1/8 of the field, 16 channels), two waves per SIMD: the product's grid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int W, int NM, bool IMG>
__global__ __launch_bounds__(256, 2) void k(char* __restrict__ dst) {
    __shared__ __align__(16) char s_img[4][16 * 272];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 4 + wv;
    const long field_bytes = 2083328, ch_bytes = 64 * field_bytes;
    const long run = wave % 8, cg = (wave / 8) % 4, f = wave / 32;
    if (f >= 64) return;
    const int tiles = 8138 / 8;
    constexpr int kLanesPerRow = 256 / W, kRowsPerInstr = 64 / kLanesPerRow, kInstr = 16 / kRowsPerInstr;
    char* base = dst + (cg * 16 + lane / kLanesPerRow) * ch_bytes + f * field_bytes + run * tiles * 256 + (lane % kLanesPerRow) * W;
    char* img = s_img[wv];
    h8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(0.01f * j + lane * 1e-3f); bv[j] = (_Float16)(0.02f * j); }
    f16v acc = {};
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (IMG) {   // the product's detour: 16 dword writes into the line image, 4 x 16-byte reads back
            const int n = lane & 31, h = lane >> 5, cl = n >> 1, comp = n & 1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float y = acc[e] * 3.7e-9f;
                reinterpret_cast<float*>(img + cl * 272)[2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + comp] = y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int i = 0; i < kInstr; ++i) {
            char* p = base + (long)i * kRowsPerInstr * ch_bytes + (long)t * 256;
            if (W == 16) {
                v4f v;
                if (IMG) v = *reinterpret_cast<const v4f*>(img + ((lane >> 4) + 4 * i) * 272 + 16 * (lane & 15));
                else v = (v4f){acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
            } else {
                const v2f v = {acc[2 * i] * 3.7e-9f, acc[2 * i + 1] * 3.7e-9f};
                __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(p));
            }
        }
    }
}

template <int W, int NM, bool IMG>
void run(char* dst, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, NM, IMG>), dim3(512), dim3(256), 0, 0, dst);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = -1; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-64s %2d MFMAs per tile: %.3f ms  %.2f TB/s\n", what, NM, best, 64.0 * 64 * 2083328 / best / 1e9);
    fflush(stdout);
}

int main() {
    char* dst;
    if (hipMalloc(&dst, 64L * 64 * 2083328 + 4096) != hipSuccess) return 1;
    run<16, 0, false>(dst, "dwordx4, 4 rows x 256 B per instruction, from registers");
    run<16, 0, true>(dst, "dwordx4, 4 rows x 256 B per instruction, through the LDS image");
    run<8, 0, false>(dst, "dwordx2, 2 rows x 256 B per instruction, from registers");
    run<16, 21, false>(dst, "dwordx4, 4 rows x 256 B per instruction, from registers");
    run<16, 21, true>(dst, "dwordx4, 4 rows x 256 B per instruction, through the LDS image");
    run<8, 21, false>(dst, "dwordx2, 2 rows x 256 B per instruction, from registers");
    run<16, 27, true>(dst, "dwordx4, 4 rows x 256 B per instruction, through the LDS image");
    run<8, 27, false>(dst, "dwordx2, 2 rows x 256 B per instruction, from registers");
    return 0;
}

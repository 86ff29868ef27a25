// Microbenchmark, second step after store_compute.hip: which ingredient of the FIR kernels stops an 8.6 GB store stream
// from overlapping with the arithmetic?  One wave handles 4 KB "tiles"; per tile it does NV packed FMAs and NM matrix
// instructions, then writes the tile with 4 coalesced 1 KB stores.  Switches:
//   LDSPATH  the tile goes through an LDS transposition (16 ds_write_b32 per lane, 4 ds_read_b128) before the stores,
//            as in the FIR kernels; otherwise it is stored straight from registers
//   BREAD    every matrix instruction takes its B operand from LDS (ds_read_b64), as the FIR kernels' symbol window does
//   BIGLDS   the block reserves 53 KB of LDS and is launched with __launch_bounds__(256, 2): 2 waves per SIMD
// "real" stores go to an 8 GB buffer, "dummy" stores to one cache-resident 1 KB line per lane set.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <bool REAL, bool LDSPATH, bool BREAD, bool BIGLDS>
__global__ __launch_bounds__(256, 2) void k(v4f* __restrict__ dst, v4f* __restrict__ dummy, long ntiles, int nv, int nm) {
    __shared__ float stage[4][16 * 65];
    __shared__ v8h bwin[4][64 * 4];
    __shared__ float pad[BIGLDS ? 8192 : 1];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * 4 + w;
    const long nwaves = (long)gridDim.x * 4;
    if (BIGLDS && ntiles < 0) pad[threadIdx.x] = 1.f;     // keeps the reservation alive
    for (int i = 0; i < 4; ++i) {
        v8h h;
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)(lane * 0.01f + j + i);
        bwin[w][lane * 4 + i] = h;
    }
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)i + lane * 1e-3f, 1.f};
    const v2f m = {0.999f, 1e-3f};
    for (long c = wave; c < ntiles; c += nwaves) {
        v16f acc = {};
        for (int it = 0; it < nm; ++it) {
            v8h bv;
            if (BREAD) bv = bwin[w][lane * 4 + (it & 3)];
            else for (int j = 0; j < 8; ++j) bv[j] = (_Float16)a[j].x;
            v8h av;
            for (int j = 0; j < 8; ++j) av[j] = (_Float16)a[j].y;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        }
        for (int it = 0; it < nv; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, a[i]);
        for (int i = 0; i < 8; ++i) { acc[2 * i] += a[i].x; acc[2 * i + 1] += a[i].y; }
        v4f o[4];
        if (LDSPATH) {
#pragma unroll
            for (int i = 0; i < 16; ++i) stage[w][i * 65 + lane] = acc[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* s = &stage[w][(lane >> 2) * 65 + (lane & 3) * 16 + r * 4];
                o[r] = (v4f){s[0], s[1], s[2], s[3]};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v4f){acc[4 * r], acc[4 * r + 1], acc[4 * r + 2], acc[4 * r + 3]};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v4f* p = REAL ? &dst[(c * 4 + r) * 64 + lane] : &dummy[lane];
            *p = o[r];
        }
        if (LDSPATH) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

template <bool REAL, bool L, bool B, bool G>
float run(v4f* dst, v4f* dummy, long ntiles, int nv, int nm, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<REAL, L, B, G>), dim3(blocks), dim3(256), 0, 0, dst, dummy, ntiles, nv, nm);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <bool L, bool B, bool G>
void row(const char* name, v4f* dst, v4f* dummy, long ntiles, int nv, int nm, int blocks) {
    const float st = run<true, L, B, G>(dst, dummy, ntiles, 0, 0, blocks);
    const float co = run<false, L, B, G>(dst, dummy, ntiles, nv, nm, blocks);
    const float bo = run<true, L, B, G>(dst, dummy, ntiles, nv, nm, blocks);
    printf("%-34s nv=%3d nm=%2d blocks=%4d  stores %.3f  compute %.3f  both %.3f ms\n", name, nv, nm, blocks, st, co, bo);
    fflush(stdout);
}

int main() {
    const long bytes = 8L << 30, ntiles = bytes / 4096;
    v4f *dst, *dummy;
    if (hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&dummy, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int blocks : {512, 2048}) {
        for (int nm : {0, 16}) {
            const int nv = nm ? 30 : 60;
            row<false, false, false>("registers", dst, dummy, ntiles, nv, nm, blocks);
            row<true, false, false>("lds path", dst, dummy, ntiles, nv, nm, blocks);
            row<true, true, false>("lds path + B from lds", dst, dummy, ntiles, nv, nm, blocks);
            row<true, true, true>("lds path + B from lds + 53 KB", dst, dummy, ntiles, nv, nm, blocks);
        }
    }
    return 0;
}

// Microbenchmark: the wide FIR's store pattern alone.  [64 channels][64 fields][260416] complex64; a wave = (field, 1/16 or
// 1/32 of the field), per 16-sample tile 32 store instructions of one dword per lane: lanes 0..31 = the 128-byte line of one
// channel row, lanes 32..63 the line of the row four further (fir_w.hip).  Compared with 8-byte and 16-byte variants that
// cover the same 64 rows x 128 B with 16 / 8 instructions (2 / 1 ... rows per half-wave), and with / without the nt bit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int W, bool NT>
__global__ __launch_bounds__(256) void k(char* __restrict__ dst, int segs) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long field_bytes = 2083328, ch_bytes = 64 * field_bytes;
    const long f = wave / segs, seg = wave % segs;
    if (f >= 64) return;
    const long seg_tiles = 16276 / segs;
    constexpr int kLanesPerRow = 128 / W;            // 32 / 16 / 8
    constexpr int kRowsPerInstr = 64 / kLanesPerRow; // 2 / 4 / 8
    char* base = dst + f * field_bytes + seg * seg_tiles * 128 + (lane % kLanesPerRow) * W;
    const long row0 = lane / kLanesPerRow;
    const v4f v = {(float)lane, 1.f, 2.f, 3.f};
    for (long t = 0; t < seg_tiles; ++t) {
#pragma unroll
        for (int i = 0; i < 64 / kRowsPerInstr; ++i) {
            // W = 4: instruction i covers rows (i & 3) + 8 ((i >> 2) & 3) + 32 (i >> 4) and that + 4
            long row;
            if (W == 4) row = (i & 3) + 8 * ((i >> 2) & 3) + 32 * (i >> 4) + 4 * row0;
            else row = (long)i * kRowsPerInstr + row0;
            char* p = base + row * ch_bytes + t * 128;
            if (W == 16) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); else *reinterpret_cast<v4f*>(p) = v; }
            else if (W == 8) { if (NT) __builtin_nontemporal_store((v2f){v.x, v.y}, reinterpret_cast<v2f*>(p)); else *reinterpret_cast<v2f*>(p) = (v2f){v.x, v.y}; }
            else { if (NT) __builtin_nontemporal_store(v.x, reinterpret_cast<float*>(p)); else *reinterpret_cast<float*>(p) = v.x; }
        }
    }
}

// the same pattern through buffer descriptors exactly as fir_w.hip issues it: eight descriptors of eight rows, per-lane
// offset 4 n + (4 h + i) * row_bytes, scalar tile offset; BURST tiles' stores back to back, then PAUSE cycles of s_sleep
template <int AUX>
__global__ __launch_bounds__(256) void kb(char* __restrict__ dst, int segs, int pause) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long field_bytes = 2083328;
    const unsigned row_bytes = 64u * 2083328u;
    const long f = wave / segs, seg = wave % segs;
    if (f >= 64) return;
    const int seg_tiles = 16276 / segs;
    char* base = dst + f * field_bytes;
    unsigned off[4];
    for (int i = 0; i < 4; ++i) off[i] = 4u * n + (unsigned)(4 * h + i) * row_bytes;
    const float v = (float)lane;
    for (int t = 0; t < seg_tiles; ++t) {
        const int soff = (int)((seg * seg_tiles + t) * 128);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)(8 * g) * row_bytes, 0, (int)(8u * row_bytes), 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off[i], soff, AUX);
        }
        for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(8);
    }
}

// the global-store pattern with NM MFMAs (and NV packed-fp16 vector instructions each) between the bursts of a tile's 32 stores;
// W = 4: dword stores, one per lane and row pair; W = 16: the same bytes as 8 dwordx4 stores (8 rows x 128 B each)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int W, int NM, int NV>
__global__ __launch_bounds__(256, 2) void km(char* __restrict__ dst, int segs) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long field_bytes = 2083328, ch_bytes = 64 * field_bytes;
    const long f = wave / segs, seg = wave % segs;
    if (f >= 64) return;
    const long seg_tiles = 16276 / segs;
    constexpr bool kLaneIsRow = W == 17;   // 16 B per lane, lane = channel row: an instruction writes 32 rows x 32 B
    constexpr int kLanesPerRow = kLaneIsRow ? 2 : 128 / W, kRowsPerInstr = 64 / kLanesPerRow;
    char* base = dst + f * field_bytes + seg * seg_tiles * 128 + (kLaneIsRow ? (lane >> 5) * 16 : (lane % kLanesPerRow) * W);
    const long row0 = kLaneIsRow ? (lane & 31) : lane / kLanesPerRow;
    h8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(0.01f * j + lane * 1e-3f); bv[j] = (_Float16)(0.02f * j); }
    f16v acc[2] = {};
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = 0x3c003c00u + i + lane;
    for (long t = 0; t < seg_tiles; ++t) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[m & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[(m * NV + i) & 7]) : "v"(0x3c013c01u + lane), "v"(0x38003800u));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < (kLaneIsRow ? 8 : 64 / kRowsPerInstr); ++i) {
            long row;
            if (W == 4) row = (i & 3) + 8 * ((i >> 2) & 3) + 32 * (i >> 4) + 4 * row0;
            else if (kLaneIsRow) row = 32 * (i >> 2) + row0;
            else row = (long)i * kRowsPerInstr + row0;
            char* p = base + row * ch_bytes + t * 128 + (kLaneIsRow ? 32 * (i & 3) : 0);
            const float x = acc[i >> 4 & 1][i & 15] + __uint_as_float(r[i & 7]);
            if (W >= 16) *reinterpret_cast<v4f*>(p) = (v4f){x, x, x, x};
            else *reinterpret_cast<float*>(p) = x;
        }
    }
}

template <int W, int NM, int NV>
void runm(char* dst, int segs) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((km<W, NM, NV>), dim3(64 * segs / 4), dim3(256), 0, 0, dst, segs);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = -1; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("global stores %2d B/lane + %2d MFMA x (1 + %d v_pk_fma_f16) per tile: segs=%2d  %.3f ms  %.2f TB/s\n", W, NM, NV, segs, best, 64.0 * 64 * 2083328 / best / 1e9);
    fflush(stdout);
}

template <int AUX>
void runb(char* dst, int segs, int pause) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kb<AUX>), dim3(64 * segs / 4), dim3(256), 0, 0, dst, segs, pause);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = -1; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("buffer stores: segs=%2d  aux=%d  pause=%2d x s_sleep 8 per tile  %.3f ms  %.2f TB/s\n", segs, AUX, pause, best, 64.0 * 64 * 2083328 / best / 1e9);
    fflush(stdout);
}

template <int W, bool NT>
void run(char* dst, int segs) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, NT>), dim3(64 * segs / 4), dim3(256), 0, 0, dst, segs);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = -1; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("segs=%2d  %2d B/lane  nt=%d  %.3f ms  %.2f TB/s\n", segs, W, (int)NT, best, 64.0 * 64 * 2083328 / best / 1e9);
    fflush(stdout);
}

int main() {
    char* dst;
    if (hipMalloc(&dst, 9L << 30) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int segs : {16, 32}) {
        runm<4, 0, 0>(dst, segs); runm<4, 14, 0>(dst, segs); runm<4, 28, 0>(dst, segs); runm<4, 28, 4>(dst, segs); runm<4, 56, 0>(dst, segs);
        runm<17, 0, 0>(dst, segs); runm<17, 28, 0>(dst, segs); runm<17, 28, 4>(dst, segs); runm<17, 56, 0>(dst, segs);
        runm<16, 0, 0>(dst, segs); runm<16, 14, 0>(dst, segs); runm<16, 28, 0>(dst, segs); runm<16, 28, 4>(dst, segs); runm<16, 56, 0>(dst, segs);
    }
    for (int segs : {32})
        for (int pause : {0, 4}) { runb<0>(dst, segs, pause); if (pause == 0) runb<2>(dst, segs, pause); }
    for (int segs : {32}) {
        run<4, true>(dst, segs); run<4, false>(dst, segs);
        run<8, true>(dst, segs); run<8, false>(dst, segs);
        run<16, true>(dst, segs); run<16, false>(dst, segs);
    }
    return 0;
}

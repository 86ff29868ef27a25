// Microbenchmark: does the ORDER in which an 8 GB buffer is written matter?  2048 waves (8 per CU), each stores 4 KB
// "tiles" (4 store instructions of 1 KB), nothing else.  Patterns:
//   interleaved   tile index = wave + k * nwaves      (at any moment the chip writes one contiguous 8 MB region)
//   streams       tile index = wave * tiles_per_wave + k  (every wave walks its own 4 MB region: 2048 write streams)
//   rows          as the FIR kernels: a tile is 32 rows x 128 B, rows 2083328 B apart, a wave walks along its 32 rows
//                 (65536 write streams of 128 B pieces)
//   product layout  rows a whole channel (133 MB, a multiple of 32 KB) apart, exactly the encoder's IQ buffer
//   rows x T      the same as rows, but T tiles (T x 128 B per row) are written back to back by one wave before it moves on
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(char* __restrict__ dst, long ntiles, int mode, int burst) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const long per = ntiles / nwaves;
    const v4f v = {(float)lane, 1.f, 2.f, 3.f};
    if (mode == 0) {
        for (long k = 0; k < per; ++k) {
            char* p = dst + (wave + k * nwaves) * 4096 + lane * 16;
            for (int r = 0; r < 4; ++r) *reinterpret_cast<v4f*>(p + r * 1024) = v;
        }
    } else if (mode == 1) {
        for (long k = 0; k < per; ++k) {
            char* p = dst + (wave * per + k) * 4096 + lane * 16;
            for (int r = 0; r < 4; ++r) *reinterpret_cast<v4f*>(p + r * 1024) = v;
        }
    } else if (mode == 3) {
        // the product layout: [64 channels][64 fields][260416] complex64; a wave = (field, 32-channel group, 1/16 of the field),
        // its 32 rows are a whole channel apart (64 x 2083328 B = a multiple of 32 KB)
        const long field_bytes = 2083328, ch_bytes = 64 * field_bytes;
        const long f = wave / 32, cg = (wave / 16) & 1, seg = wave & 15;
        if (f >= 64) return;
        const long seg_tiles = 1017;
        char* base = dst + (cg * 32 + (lane >> 3)) * ch_bytes + f * field_bytes + seg * seg_tiles * 128 + (lane & 7) * 16;
        for (long k = 0; k < seg_tiles; ++k)
            for (int r = 0; r < 4; ++r) *reinterpret_cast<v4f*>(base + (long)r * 8 * ch_bytes + k * 128) = v;
    } else {
        // rows: the buffer is [rows_total][row_bytes]; 32 rows per wave-group, `segs` waves share a row set, each its own
        // segment of the row
        const long row_bytes = 2083328;
        const long rows_total = (ntiles * 4096) / row_bytes / 32 * 32;
        const long groups = rows_total / 32;
        const long segs = nwaves / groups;                       // waves per row set
        if (segs == 0) return;
        const long grp = wave / segs, seg = wave % segs;
        if (grp >= groups) return;
        const long seg_tiles = row_bytes / 128 / segs;
        char* base = dst + (grp * 32 + (lane >> 3)) * row_bytes + seg * seg_tiles * 128 + (lane & 7) * 16;
        for (long k = 0; k + burst <= seg_tiles; k += burst)
            for (int b = 0; b < burst; ++b)
                for (int r = 0; r < 4; ++r) *reinterpret_cast<v4f*>(base + (long)r * 8 * row_bytes + (k + b) * 128) = v;
    }
}

int main() {
    const long bytes = 8L << 30, ntiles = bytes / 4096;   // 8 GiB >= 64 x 64 x 2083328 B: every pattern stays inside
    static_assert((8L << 30) >= 64L * 64 * 2083328, "the product layout must fit the buffer");
    char* dst;
    if (hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"interleaved", "streams (4 KB pieces)", "rows (128 B pieces)", "rows x 2", "rows x 4", "rows x 8", "rows x 16", "product layout"};
    const int modes[] = {0, 1, 2, 2, 2, 2, 2, 3}, bursts[] = {1, 1, 1, 2, 4, 8, 16, 1};
    for (int blocks : {512, 1024}) {
        for (int i = 0; i < 8; ++i) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dst, ntiles, modes[i], bursts[i]);
                hipError_t le = hipGetLastError();
                (void)hipEventRecord(e1); hipError_t se = hipEventSynchronize(e1);
                float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, e0, e1);
                if (le != hipSuccess || se != hipSuccess || ee != hipSuccess) { printf("launch %s / sync %s / elapsed %s\n", hipGetErrorString(le), hipGetErrorString(se), hipGetErrorString(ee)); return 1; }
                if (ms < best) best = ms;
            }
            printf("blocks=%4d  %-24s %.3f ms\n", blocks, names[i], best);
            fflush(stdout);
        }
    }
    return 0;
}

// tools/fetch_calib.hip — what does rocprofv3's FETCH_SIZE report for THIS library's access patterns?
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE on wide coalesced streams only (it reports half their bytes on gfx950) and says
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The accumulation kernels
// (csrc/kernels_msm.cuh k_msm_accum) gather ONE packed point per lane and step: 64 bytes (BN254 G1: four 16-byte loads of one
// lane), 128 bytes (G2), at addresses a sorted list makes random.  This program issues exactly those patterns over tables far
// larger than the 256 MiB Infinity Cache, with byte counts known by construction; run it under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/fetch_calib
// and divide: factor(pattern) = bytes requested / FETCH_SIZE.  tools/pmc_traffic.py applies the factor of the pattern a kernel
// uses (stored in profiles/pmc_traffic.json under "calibration").   (development probe; built by tools/gpu_r6*.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64; typedef unsigned u32;

// every lane reads `per_lane` consecutive uint4 of a coalesced stream (16 B per lane and load: the guide's calibrated pattern)
__global__ void k_stream16(const uint4* __restrict__ src, u64 n16, u32* __restrict__ sink) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (; i < n16; i += stride) { const uint4 t = src[i]; acc ^= t.x ^ t.y ^ t.z ^ t.w; }
    if (acc == 0x12345u) sink[0] = acc;
}
// 4 bytes per lane, coalesced (how the sorted list and the digits are read)
__global__ void k_stream4(const u32* __restrict__ src, u64 n, u32* __restrict__ sink) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (; i < n; i += stride) acc ^= src[i];
    if (acc == 0x12345u) sink[0] = acc;
}
// one lane gathers NQ consecutive uint4 (NQ * 16 bytes, aligned to their own size) at a pseudo-random record of the table:
// NQ = 4 is aff_load_words of a BN254 G1 point, NQ = 8 of a G2 point, NQ = 6 / 12 the BLS12-381 ones
template <int NQ>
__global__ void k_gather(const uint4* __restrict__ tbl, u64 nrec_mask, u64 count, u32* __restrict__ sink) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (; i < count; i += stride) {
        u64 h = i * 0x9E3779B97F4A7C15ull;          // a bijection of the index: every record at most once per sweep of 2^k indices
        h ^= h >> 29;
        const uint4* p = tbl + (h & nrec_mask) * NQ;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const uint4 t = p[q]; acc ^= t.x ^ t.y ^ t.z ^ t.w; }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
// the rows pass of the NTT (csrc/kernels_ntt.cuh ntt_load): lane i reads the two uint4 of element i — 16 B at a 32-byte stride per
// instruction, two instructions per element, over a contiguous run
__global__ void k_pair16(const uint4* __restrict__ src, u64 nelem, u32* __restrict__ sink) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (; i < nelem; i += stride) { const uint4 lo = src[2 * i], hi = src[2 * i + 1]; acc ^= lo.x ^ lo.w ^ hi.y ^ hi.z; }
    if (acc == 0x12345u) sink[0] = acc;
}
// the cols pass: C adjacent 32-byte elements per row (64 bytes at C = 2, 128 at C = 4), rows `row_elems` elements apart; a workgroup
// walks the rows of its C columns, every lane reading the two uint4 of one element
template <int C>
__global__ void k_segments(const uint4* __restrict__ src, u64 row_elems, u64 rows, u32* __restrict__ sink) {
    const u64 c0 = (u64)blockIdx.x * C;                 // first column of this workgroup
    u32 acc = 0;
    for (u64 e = threadIdx.x; e < rows * C; e += blockDim.x) {
        const u64 a = e / C, j = e % C, g = a * row_elems + c0 + j;
        const uint4 lo = src[2 * g], hi = src[2 * g + 1];
        acc ^= lo.x ^ lo.w ^ hi.y ^ hi.z;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ void k_fill(uint4* p, u64 n16) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) p[i] = make_uint4((u32)i, (u32)(i >> 7), 0x9e3779b9u * (u32)i, 7u);
}
template <class F> float time_it(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const u64 table_bytes = (u64)8 << 30;           // 8 GiB: 32x the Infinity Cache
    const u64 n16 = table_bytes / 16;
    uint4* tbl; u32* sink;
    CK(hipMalloc(&tbl, table_bytes)); CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tbl, n16);
    CK(hipDeviceSynchronize());
    const dim3 G(256 * 8), T(256);
    const u64 want = (u64)2 << 30;                  // bytes each pattern requests
    for (int rep = 0; rep < 2; ++rep) {
        float ms = time_it([&] { hipLaunchKernelGGL(k_stream16, G, T, 0, 0, (const uint4*)tbl, want / 16, sink); });
        printf("stream16   requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_stream4, G, T, 0, 0, (const u32*)tbl + (n16 * 2), want / 4, sink); });
        printf("stream4    requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_gather<4>, G, T, 0, 0, (const uint4*)tbl, table_bytes / 64 - 1, want / 64, sink); });
        printf("gather64   requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_gather<8>, G, T, 0, 0, (const uint4*)tbl, table_bytes / 128 - 1, want / 128, sink); });
        printf("gather128  requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_gather<2>, G, T, 0, 0, (const uint4*)tbl, table_bytes / 32 - 1, want / 32, sink); });
        printf("gather32   requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(k_pair16, G, T, 0, 0, (const uint4*)tbl, want / 32, sink); });
        printf("pair16     requested %llu bytes  %.3f ms  %.0f GB/s\n", want, ms, want / ms / 1e6);
        {   // a 2^13 x 2^13 matrix of 32-byte elements (2 GiB): every column group once
            const u64 n = 1 << 13;
            ms = time_it([&] { hipLaunchKernelGGL(k_segments<2>, dim3((unsigned)(n / 2)), dim3(512), 0, 0, (const uint4*)tbl, n, n, sink); });
            printf("seg64      requested %llu bytes  %.3f ms  %.0f GB/s\n", n * n * 32, ms, n * n * 32 / ms / 1e6);
            ms = time_it([&] { hipLaunchKernelGGL(k_segments<4>, dim3((unsigned)(n / 4)), dim3(512), 0, 0, (const uint4*)tbl, n, n, sink); });
            printf("seg128     requested %llu bytes  %.3f ms  %.0f GB/s\n", n * n * 32, ms, n * n * 32 / ms / 1e6);
        }
    }
    return 0;
}

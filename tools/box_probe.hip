// tools/box_probe.hip — why do some boxes run the latency-bound kernels 2x slower?  Times (a) a lone wavefront's
// dependent multiply-add chain (effective clock seen by a latency-bound kernel), (b) the same chain with the GPU full,
// (c) a dependent LDS read chain, (d) a dependent global-load chain through L2.   (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64; typedef unsigned u32;
__global__ void k_chain(u32* out, u32 a, int iters) {
    u64 acc = threadIdx.x;
    u32 x = a + threadIdx.x;
    for (int i = 0; i < iters; ++i) { acc = (u64)x * (u32)acc + acc; x ^= (u32)(acc >> 32); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc;
}
__global__ void k_lds_chain(u32* out, int iters) {
    __shared__ u32 t[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) t[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    u32 p = threadIdx.x;
    for (int i = 0; i < iters; ++i) p = t[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
}
__global__ void k_mem_chain(u32* out, const u32* tbl, u32 mask, int iters) {
    u32 p = threadIdx.x * 977 & mask;
    for (int i = 0; i < iters; ++i) p = tbl[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
}
template <class F> float time_it(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    u32* out; CK(hipMalloc(&out, 4 << 20));
    const int it = 200000;
    float lone = time_it([&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, 12345u, it); });
    float full = time_it([&] { hipLaunchKernelGGL(k_chain, dim3(2048), dim3(256), 0, 0, out, 12345u, it); });
    printf("mad chain: lone wave %.3f ms (%.1f ns/iter), full GPU %.3f ms\n", lone, lone * 1e6 / it, full);
    float lds = time_it([&] { hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 0, 0, out, it); });
    printf("lds chain: lone wave %.3f ms (%.1f ns/hop)\n", lds, lds * 1e6 / it);
    for (u32 lg : {16u, 22u, 26u}) {
        const u32 n = 1u << lg;
        std::vector<u32> h(n);
        u32 x = 1; for (u32 i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x >> 4) & (n - 1); }
        u32* tbl; CK(hipMalloc(&tbl, (size_t)n * 4)); CK(hipMemcpy(tbl, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        float t = time_it([&] { hipLaunchKernelGGL(k_mem_chain, dim3(1), dim3(64), 0, 0, out, tbl, n - 1, 20000); });
        printf("global chain over %4u KiB: %.1f ns/hop\n", n * 4 / 1024, t * 1e6 / 20000);
        CK(hipFree(tbl));
    }
    return 0;
}

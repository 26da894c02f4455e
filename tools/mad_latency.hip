// tools/mad_latency.hip — how many independent v_mad_u64_u32 chains a SIMD of gfx950 needs in flight to issue one every 4 cycles
// (development probe, not part of libzkhip): chains per wave x waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mad_latency.hip -o tools/mad_latency && tools/mad_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned u32;
#define M1(acc) "v_mad_u64_u32 %" #acc ", s[10:11], %4, %5, %" #acc "\n"
#define R8(x) x x x x x x x x
template <int CH>
__global__ void __launch_bounds__(64) k(u64* out, u32 a, u32 b, int iters) {
    u64 c0 = threadIdx.x, c1 = threadIdx.x + 1, c2 = threadIdx.x + 2, c3 = threadIdx.x + 3;
    u32 x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
        if (CH == 1) asm volatile(R8(R8(M1(0))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s10", "s11");
        if (CH == 2) asm volatile(R8(R8(M1(0) M1(1))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s10", "s11");
        if (CH == 4) asm volatile(R8(R8(M1(0) M1(1) M1(2) M1(3))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s10", "s11");
        if (CH == 11) asm volatile(R8(R8(M1(0) "s_nop 0\n")) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s10", "s11");   // one chain, s_nop after each
        if (CH == 12) asm volatile(R8(R8(M1(0) "s_nop 0\n" M1(1) "s_nop 0\n")) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(y) : "s10", "s11");
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3;
}
template <int CH>
static void run(int waves_per_simd, int madsper) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 2000;
    u64* d; CK(hipMalloc(&d, (size_t)blocks * 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mads_per_simd = (double)waves_per_simd * iters * 64.0 * madsper;      // wave-instructions per SIMD
    printf("chains %2d  waves/SIMD %d: %.3f ms, %.2f ns per wave-MAD per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", CH, waves_per_simd, ms, ms * 1e6 / mads_per_simd);
    CK(hipFree(d));
}
int main() {
    for (int w = 1; w <= 4; ++w) { run<1>(w, 1); run<2>(w, 2); run<4>(w, 4); run<11>(w, 1); run<12>(w, 2); }
    return 0;
}

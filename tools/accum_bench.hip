// tools/accum_bench.hip — the bucket-accumulation kernel in isolation on synthetic sorted lists (development probe):
// register-budget (launch-bounds) variants x points-per-lane, for G1 and G2 of BN254.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/accum_bench.hip -o tools/accum_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/kernels_msm.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F, int WPE>
float run(const Aff<F>* bases, const u32* off, const u32* sorted, u32* lane_key, Xyzz<F>* partial, u32 nkeys, u32 P, u32 total) {
    const u32 nlanes = (total + P - 1) / P;
    hipLaunchKernelGGL(k_msm_lane_keys, dim3((nlanes + 255) / 256), dim3(256), 0, 0, off, nkeys, P, nlanes, lane_key);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_msm_accum<F, WPE>), dim3((nlanes + 255) / 256), dim3(256), 0, 0, bases, off, sorted, lane_key, partial, nkeys, P, nlanes);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}
template <class F>
void bench(const char* name, u32 npts, u32 nkeys, u32 per_bucket) {
    const u32 total = nkeys * per_bucket;
    std::vector<u32> hb((size_t)npts * sizeof(Aff<F>) / 4);
    for (auto& v : hb) v = ((u32)rand() * 2654435761u) & 0x000fffffu;   // small limbs: valid TIGHT operands for either field type
    std::vector<u32> hoff(nkeys + 1), hs(total);
    for (u32 k = 0; k <= nkeys; ++k) hoff[k] = k * per_bucket;
    for (auto& v : hs) v = (((u32)rand() << 12) ^ (u32)rand()) % npts | ((rand() & 1) << 31);
    Aff<F>* bases; u32 *off, *sorted, *lane_key; Xyzz<F>* partial;
    CK(hipMalloc(&bases, hb.size() * 4)); CK(hipMalloc(&off, hoff.size() * 4)); CK(hipMalloc(&sorted, hs.size() * 4));
    CK(hipMalloc(&lane_key, (size_t)total * 4)); CK(hipMalloc(&partial, ((size_t)nkeys + total / 8 + 8) * sizeof(Xyzz<F>)));
    CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(off, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sorted, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    for (u32 P : {16u, 32u, 64u}) {
        float t1 = run<F, 1>(bases, off, sorted, lane_key, partial, nkeys, P, total);
        float t2 = run<F, 2>(bases, off, sorted, lane_key, partial, nkeys, P, total);
        float t3 = run<F, 3>(bases, off, sorted, lane_key, partial, nkeys, P, total);
        float t4 = run<F, 4>(bases, off, sorted, lane_key, partial, nkeys, P, total);
        printf("%s total=%u P=%2u | WPE1 %8.3f ms (%6.2f Gmadd/s) | WPE2 %8.3f (%6.2f) | WPE3 %8.3f (%6.2f) | WPE4 %8.3f (%6.2f)\n", name, total, P, t1,
               total / t1 * 1e-6, t2, total / t2 * 1e-6, t3, total / t3 * 1e-6, t4, total / t4 * 1e-6);
    }
    CK(hipFree(bases)); CK(hipFree(off)); CK(hipFree(sorted)); CK(hipFree(lane_key)); CK(hipFree(partial));
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    bench<Fu<Bn254Fq>>("G1 unsat", 1u << 20, 1u << 19, 32);
    bench<Fu2<Bn254Fq>>("G2 unsat", 1u << 20, 1u << 19, 32);

    return 0;
}

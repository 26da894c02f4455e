// tools/accum_bench.hip — the bucket-accumulation kernel in isolation on synthetic sorted lists (development probe; the
// "peak" of bench.py's compute_bound figure): 2^15 equally full buckets, random entries into a 16-level table of 2^20
// random packed points, one slice per resident work-item, for 1..5 tables per launch (the proof runs A, B1, L as one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/accum_bench.hip -o tools/accum_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/kernels_msm.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F, int WPE>
float run(const MsmTables& tb, int nt, const u32* off, const u32* sorted, u32* lane_key, Xyzz<F>* partial, u32 nkeys, u32 nlanes) {
    const MsmCut cut{nlanes, 1, 0x7fffffffu};
    hipLaunchKernelGGL(k_msm_lane_keys, dim3((nlanes + 255) / 256), dim3(256), 0, 0, off, nkeys, cut, lane_key);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    constexpr size_t acc_lds = msm_accum_lds_bytes<F>();
    if (acc_lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k_msm_accum<F, WPE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_msm_accum<F, WPE, false>), dim3((nlanes + 255) / 256, nt), dim3(256), acc_lds, 0, tb, off, sorted, lane_key, partial, (u64)nkeys + nlanes, nkeys, cut);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}
template <class F, int WPE>
void bench(const char* name, int cus) {
    const u32 npts = 1u << 20, levels = 16, nkeys = 1u << 15, per_bucket = 512, total = nkeys * per_bucket;   // = 2^20 x 16 digits
    constexpr int NW = AffPacked<F>::NW;
    std::vector<u32> hb((size_t)npts * levels * 2 * NW);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (i % NW == NW - 1) ? ((u32)rand() & 0x0fffffffu) : ((u32)rand() * 2654435761u);   // coordinates < p
    std::vector<u32> hoff(nkeys + 1), hs(total);
    for (u32 k = 0; k <= nkeys; ++k) hoff[k] = k * per_bucket;
    for (auto& v : hs) v = (u32)(((((u64)rand() << 16) ^ (u64)rand()) % ((u64)npts * levels))) | ((u32)(rand() & 1) << 31);
    const int max_t = 3;
    AffPacked<F>* bases[max_t]; u32 *off, *sorted, *lane_key; Xyzz<F>* partial;
    MsmTables tb{};
    for (int t = 0; t < max_t; ++t) {
        CK(hipMalloc(&bases[t], hb.size() * 4));
        CK(hipMemcpy(bases[t], hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        tb.p[t] = bases[t];
    }
    const u32 max_lanes = (u32)cus * 4 * 64 * 8;
    CK(hipMalloc(&off, hoff.size() * 4)); CK(hipMalloc(&sorted, hs.size() * 4));
    CK(hipMalloc(&lane_key, (size_t)max_lanes * 4)); CK(hipMalloc(&partial, (size_t)max_t * ((size_t)nkeys + max_lanes) * sizeof(Xyzz<F>)));
    CK(hipMemcpy(off, hoff.data(), hoff.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sorted, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    for (int nt : {1, 3})
        for (int waves : {2, 3, 4, 5, 6}) {
            const u32 nlanes = (u32)cus * 4 * 64 * waves / nt;
            const float ms = run<F, WPE>(tb, nt, off, sorted, lane_key, partial, nkeys, nlanes);
            printf("%s tables=%d waves/SIMD=%d lanes/table=%6u | %8.3f ms | %6.2f G mixed additions/s\n", name, nt, waves, nlanes, ms, (double)total * nt / ms * 1e-6);
        }
    for (int t = 0; t < max_t; ++t) CK(hipFree(bases[t]));
    CK(hipFree(off)); CK(hipFree(sorted)); CK(hipFree(lane_key)); CK(hipFree(partial));
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    bench<Fu<Bn254Fq>, MsmTuning<Fu<Bn254Fq>>::ACCUM_WPE>("G1", prop.multiProcessorCount);
    bench<Fu2<Bn254Fq>, MsmTuning<Fu2<Bn254Fq>>::ACCUM_WPE>("G2", prop.multiProcessorCount);
    return 0;
}

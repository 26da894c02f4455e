// tools/microbench.hip — instruction-rate / memory-primitive probes that decide the field-multiplier and
// bucket-sort designs (not part of the product library).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/ec.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
// 8 independent chains per lane
__global__ void k_mad64(u32* out, u32 a, u32 b) {
    u64 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k;
    u32 x = a + threadIdx.x, y = b;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = (u64)x * y + acc[k];
        x ^= (u32)acc[0];
    }
    u64 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
__global__ void k_mullo(u32* out, u32 a, u32 b) {
    u32 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k + a;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] * (b + k) ;
    }
    u32 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(u32* out, u32 a, u32 b) {
    u32 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k + a;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __umulhi(acc[k], b + k) + 0x9e3779b9u;
    }
    u32 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(u32* out, u32 a, u32 b) {
    u32 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k + a;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __umul24(acc[k], b + k) + acc[(k + 1) & 7];
    }
    u32 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(u32* out, u32 a, u32 b) {
    u64 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k + a;
    u64 y = ((u64)b << 32) | a;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] + (y ^ acc[(k + 1) & 7]);
    }
    u64 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
__global__ void k_add32(u32* out, u32 a, u32 b) {
    u32 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k + a;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] + (b ^ acc[(k + 1) & 7]);
    }
    u32 s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(u32* out, double a, double b) {
    double acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_fma(acc[k], a, b);
    }
    double s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}
__global__ void k_fma32(u32* out, float a, float b) {
    float acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = threadIdx.x + k;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_fmaf(acc[k], a, b);
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}
typedef Fe<Bn254Fq> Fq;
__global__ void k_femul(Fq* out, const Fq* in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = in[i], y = in[i + 1];
    for (int k = 0; k < iters; ++k) { x = fe_mul(x, y); y = fe_mul(y, x); }
    out[i] = fe_add(x, y);
}
__global__ void k_feadd(Fq* out, const Fq* in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = in[i], y = in[i + 1];
    for (int k = 0; k < iters; ++k) { x = fe_add(x, y); y = fe_sub(y, x); }
    out[i] = fe_add(x, y);
}
__global__ void k_madd(Xyzz<Fq>* out, const Aff<Fq>* pts, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz<Fq> acc = Xyzz<Fq>::from_affine(pts[i]);
    Aff<Fq> p = pts[i + 1];
    for (int k = 0; k < iters; ++k) { acc = xyzz_madd(acc, p); p.x = acc.x; }   // p is not a curve point; arithmetic cost is what matters
    out[i] = acc;
}
// histogram with global atomics (no return) / with return
__global__ void k_hist(const u32* keys, u32* bins, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&bins[keys[i]], 1u);
}
__global__ void k_hist_ret(const u32* keys, u32* bins, u32* pos, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[i] = atomicAdd(&bins[keys[i]], 1u);
}
// random 64-B gather
struct P64 { uint4 a, b, c, d; };
__global__ void k_gather(const P64* src, const u32* idx, uint4* out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    P64 p = src[idx[i]];
    uint4 r; r.x = p.a.x ^ p.b.x ^ p.c.x ^ p.d.x; r.y = p.a.y ^ p.d.y; r.z = p.b.z ^ p.c.z; r.w = p.a.w ^ p.d.w;
    out[i] = r;
}

template <class F> float time_it(F f, int reps = 3) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s %s CUs=%d clock=%d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    const double lanes = (double)blocks * threads;
    u32* out; CK(hipMalloc(&out, lanes * 4));
    auto report = [&](const char* name, float ms, double ops_per_lane) {
        double total = lanes * ops_per_lane;
        printf("%-14s %8.3f ms  %8.2f Gop/s  (%.2f op/clk/CU @2.4GHz)\n", name, ms, total / ms * 1e-6, total / (ms * 1e-3) / (prop.multiProcessorCount * 2.4e9));
    };
    report("mad_u64_u32", time_it([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("mul_lo_u32", time_it([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("mul_hi_u32+add", time_it([&] { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("mul_u24+add", time_it([&] { hipLaunchKernelGGL(k_mad24, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("add_u64(+xor)", time_it([&] { hipLaunchKernelGGL(k_add64, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("add_u32(+xor)", time_it([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 678u); }), 8.0 * ITERS);
    report("fma_f64", time_it([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001, 1e-9); }), 8.0 * ITERS);
    report("fma_f32", time_it([&] { hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(threads), 0, 0, out, 1.0000001f, 1e-9f); }), 8.0 * ITERS);
    {
        Fq *in, *o; CK(hipMalloc(&in, (size_t)(2 * lanes + 4) * sizeof(Fq))); CK(hipMalloc(&o, lanes * sizeof(Fq)));
        std::vector<u32> h((size_t)(2 * lanes + 4) * 8);
        for (auto& v : h) v = rand() * 2654435761u; for (size_t i = 7; i < h.size(); i += 8) h[i] &= 0x1fffffff;
        CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const int it = 512;
        report("fe_mul bn254", time_it([&] { hipLaunchKernelGGL(k_femul, dim3(blocks), dim3(threads), 0, 0, o, in, it); }), 2.0 * it);
        report("fe_add/sub", time_it([&] { hipLaunchKernelGGL(k_feadd, dim3(blocks), dim3(threads), 0, 0, o, in, it); }), 2.0 * it);
        Xyzz<Fq>* xo; CK(hipMalloc(&xo, lanes * sizeof(Xyzz<Fq>)));
        const int it2 = 128;
        for (int bt : {64, 128, 256}) {
            int bl = (int)(lanes / bt);
            float ms = time_it([&] { hipLaunchKernelGGL(k_madd, dim3(bl), dim3(bt), 0, 0, xo, (const Aff<Fq>*)in, it2); });
            printf("xyzz_madd (block %3d) %8.3f ms  %8.2f Gmadd/s\n", bt, ms, lanes * it2 / ms * 1e-6);
        }
    }
    {
        const u64 n = 16u << 20; const u32 nb = 1u << 19;
        std::vector<u32> keys(n); for (auto& k : keys) k = (u32)(((u64)rand() * 2654435761u) >> 7) % nb;
        u32 *dk, *db, *dp; CK(hipMalloc(&dk, n * 4)); CK(hipMalloc(&db, nb * 4)); CK(hipMalloc(&dp, n * 4));
        CK(hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(db, 0, nb * 4));
        float ms = time_it([&] { hipLaunchKernelGGL(k_hist, dim3(n / 256), dim3(256), 0, 0, dk, db, n); });
        printf("atomic hist   16M keys / 512K bins: %.3f ms  (%.1f Gatom/s)\n", ms, n / ms * 1e-6);
        ms = time_it([&] { hipLaunchKernelGGL(k_hist_ret, dim3(n / 256), dim3(256), 0, 0, dk, db, dp, n); });
        printf("atomic hist+ret                    : %.3f ms  (%.1f Gatom/s)\n", ms, n / ms * 1e-6);
        // skewed: half of the keys hit bin 0
        for (u64 i = 0; i < n; i += 2) keys[i] = 0;
        CK(hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice));
        ms = time_it([&] { hipLaunchKernelGGL(k_hist_ret, dim3(n / 256), dim3(256), 0, 0, dk, db, dp, n); });
        printf("atomic hist+ret, 50%% on one bin    : %.3f ms\n", ms);
        P64* src; uint4* go; const u64 npts = 1u << 20;
        CK(hipMalloc(&src, npts * 64)); CK(hipMalloc(&go, n * 16));
        for (auto& k : keys) k = (u32)(((u64)rand() * 2654435761u) >> 7) % npts;
        CK(hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice));
        ms = time_it([&] { hipLaunchKernelGGL(k_gather, dim3(n / 256), dim3(256), 0, 0, src, dk, go, n); });
        printf("random 64-B gather 16M from 64 MiB : %.3f ms  (%.1f GB/s)\n", ms, n * 64.0 / ms * 1e-6);
    }
    return 0;
}

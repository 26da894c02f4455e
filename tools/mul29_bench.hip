// tools/mul29_bench.hip — variants of the 9 x 29-bit lazy Comba Montgomery product (fieldu.cuh fu_mul_inl) measured in
// isolation on gfx950 (development probe, not part of libzkhip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mul29_bench.hip -o tools/mul29_bench && tools/mul29_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/fieldu.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef Bn254Fq P;
typedef Fu<P> U;
typedef UConst<P> C;
static constexpr int N = 9, B = 29;
static constexpr u32 M = (1u << 29) - 1;

// V1: one accumulator chain (the carry of column k is the addend the next column starts from: no separate add)
__device__ __forceinline__ U mul_v1(const U& a, const U& b) {
    u32 m[N];
    U r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * C::p(k - i);
        }
        acc += (u64)a.v[k] * b.v[0];
        m[k] = ((u32)acc * C::NINV) & M;
        acc += (u64)m[k] * C::p(0);
        acc >>= B;
        asm volatile("" : "+v"(acc));
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * C::p(k - i);
        }
        r.v[k - N] = (u32)acc & M;
        acc >>= B;
        asm volatile("" : "+v"(acc));
    }
    r.v[N - 1] = (u32)acc;
    return r;
}
// V2: m_k through v_mad_u64_u32 instead of v_mul_lo_u32
__device__ __forceinline__ u32 lo_mul_mad(u32 x, u32 k) {
    u64 t;
    asm("v_mad_u64_u32 %0, s[100:101], %1, %2, 0" : "=v"(t) : "v"(x), "v"(k) : "s100", "s101");
    return (u32)t;
}
template <bool CHAIN>
__device__ __forceinline__ U mul_v2(const U& a, const U& b) {
    u32 m[N];
    U r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * C::p(k - i);
        }
        acc += (u64)a.v[k] * b.v[0];
        m[k] = lo_mul_mad((u32)acc, C::NINV) & M;
        acc += (u64)m[k] * C::p(0);
        acc >>= B;
        if (CHAIN) asm volatile("" : "+v"(acc));
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * C::p(k - i);
        }
        r.v[k - N] = (u32)acc & M;
        acc >>= B;
        if (CHAIN) asm volatile("" : "+v"(acc));
    }
    r.v[N - 1] = (u32)acc;
    return r;
}
// V4: the a*b and m*p partial products in two accumulators (independent chains inside one product)
__device__ __forceinline__ U mul_v4(const U& a, const U& b) {
    u32 m[N];
    U r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        u64 q = 0;
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            q += (u64)m[i] * C::p(k - i);
        }
        acc += (u64)a.v[k] * b.v[0];
        acc += q;
        m[k] = ((u32)acc * C::NINV) & M;
        acc += (u64)m[k] * C::p(0);
        acc >>= B;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        u64 q = 0;
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            q += (u64)m[i] * C::p(k - i);
        }
        acc += q;
        r.v[k - N] = (u32)acc & M;
        acc >>= B;
    }
    r.v[N - 1] = (u32)acc;
    return r;
}

// V5/V6/V7: NACC partial accumulators per column (round-robin over the partial products), optional chain barrier
template <int NACC, bool CHAIN>
__device__ __forceinline__ U mul_vn(const U& a, const U& b) {
    u32 m[N];
    U r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
        u64 q[NACC];
#pragma unroll
        for (int t = 0; t < NACC; ++t) q[t] = 0;
        int slot = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (k - i >= 0 && k - i < N) {
                if (!(k < N && i == k)) {    // the a_k * b_0 term of the low half goes last (after it m_k is known)
                    q[slot % NACC] += (u64)a.v[i] * b.v[k - i];
                    ++slot;
                }
                if (i < k && i < N && k - i < N && (k < N ? i < k : true)) {
                    q[slot % NACC] += (u64)m[i] * C::p(k - i);
                    ++slot;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc += q[t];
        if (k < N) {
            acc += (u64)a.v[k] * b.v[0];
            m[k] = ((u32)acc * C::NINV) & M;
            acc += (u64)m[k] * C::p(0);
        } else {
            r.v[k - N] = (u32)acc & M;
        }
        acc >>= B;
        if (CHAIN) asm volatile("" : "+v"(acc));
    }
    r.v[N - 1] = (u32)acc;
    return r;
}

template <int V>
__device__ __forceinline__ U mulv(const U& a, const U& b) {
    if (V == 5) return mul_vn<3, false>(a, b);
    if (V == 6) return mul_vn<2, true>(a, b);
    if (V == 7) return mul_vn<4, false>(a, b);
    if (V == 0) return fu_mul_inl(a, b);
    if (V == 1) return mul_v1(a, b);
    if (V == 2) return mul_v2<false>(a, b);
    if (V == 3) return mul_v2<true>(a, b);
    return mul_v4(a, b);
}
template <int V, int WPE>
__global__ void __launch_bounds__(256, WPE) k_bench(U* out, const U* in, int iters) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    U x = in[g], y = in[g + 1], z = in[g + 2], w = in[g + 3];
    for (int i = 0; i < iters; ++i) {      // two independent chains, as the curve formulas offer
        x = mulv<V>(x, y);
        z = mulv<V>(z, w);
        y = mulv<V>(y, x);
        w = mulv<V>(w, z);
    }
    out[g] = fe_add(fe_add(x, y), fe_add(z, w));
}
template <int V, int WPE>
static void run(const char* name, U* d_out, const U* d_in, int cus, std::vector<U>& ref) {
    const int blocks = cus * WPE, threads = 256, iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bench<V, WPE>), dim3(blocks), dim3(threads), 0, 0, d_out, d_in, 10);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_bench<V, WPE>), dim3(blocks), dim3(threads), 0, 0, d_out, d_in, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<U> h(1024);
    CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(U), hipMemcpyDeviceToHost));
    int bad = 0;
    if (ref.empty()) ref = h;
    else
        for (size_t i = 0; i < h.size(); ++i) {   // compare modulo p through the saturated form
            Fe<P> a = fu_to_fe(h[i]), b = fu_to_fe(ref[i]);
            if (!a.equals(b)) ++bad;
        }
    const double muls = (double)blocks * threads * iters * 4;
    printf("%-44s WPE %d: %7.3f ms  %7.1f Gmul/s  mismatches vs V0: %d\n", name, WPE, best, muls / best * 1e-6, bad);
}
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t n = (size_t)cus * 4 * 256 + 8;
    std::vector<U> h(n);
    srand(1);
    for (auto& x : h) { for (int i = 0; i < 9; ++i) x.v[i] = ((u32)rand() * 2654435761u) & M; x.v[8] &= 0x3fffff; }
    U *d_in, *d_out;
    CK(hipMalloc(&d_in, n * sizeof(U))); CK(hipMalloc(&d_out, n * sizeof(U)));
    CK(hipMemcpy(d_in, h.data(), n * sizeof(U), hipMemcpyHostToDevice));
    std::vector<U> ref3, ref2;
    run<0, 3>("V0 fu_mul_inl as compiled", d_out, d_in, cus, ref3);
    run<1, 3>("V1 single accumulator chain", d_out, d_in, cus, ref3);
    run<2, 3>("V2 m_k via v_mad_u64_u32", d_out, d_in, cus, ref3);
    run<3, 3>("V3 chain + m_k via mad", d_out, d_in, cus, ref3);
    run<4, 3>("V4 a*b and m*p in two accumulators", d_out, d_in, cus, ref3);
    run<5, 3>("V5 three partial accumulators per column", d_out, d_in, cus, ref3);
    run<6, 3>("V6 two partial accumulators + chain", d_out, d_in, cus, ref3);
    run<7, 3>("V7 four partial accumulators per column", d_out, d_in, cus, ref3);
    run<0, 2>("V0 fu_mul_inl as compiled", d_out, d_in, cus, ref2);
    run<5, 2>("V5 three partial accumulators per column", d_out, d_in, cus, ref2);
    run<7, 2>("V7 four partial accumulators per column", d_out, d_in, cus, ref2);
    run<1, 2>("V1 single accumulator chain", d_out, d_in, cus, ref2);
    run<3, 2>("V3 chain + m_k via mad", d_out, d_in, cus, ref2);
    run<4, 2>("V4 a*b and m*p in two accumulators", d_out, d_in, cus, ref2);
    return 0;
}

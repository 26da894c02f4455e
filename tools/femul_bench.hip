// tools/femul_bench.hip — candidate 254-bit Montgomery multipliers for gfx950, measured in isolation
// (development probe, not part of libzkhip).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/femul_bench.hip -o tools/femul_bench
// Prints throughput per variant and a few input/output triples (hex) that tools/femul_check.py verifies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/field.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef Bn254Fq P;
typedef Fe<P> Fq;

// ---------------- variant B: 9 x 29-bit limbs, Comba (product scanning), lazy: no carries in the inner loop
struct F29 { u32 v[9]; };
static constexpr u32 M29 = (1u << 29) - 1;
struct P29 {   // p in 29-bit limbs and -p^-1 mod 2^29, computed on the host
    u32 p[9];
    u32 inv;
};
__constant__ P29 c_p29;

__device__ __forceinline__ F29 mul29(const F29& a, const F29& b) {
    u32 m[9];
    F29 r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * c_p29.p[k - i];
        }
        acc += (u64)a.v[k] * b.v[0];
        m[k] = ((u32)acc * c_p29.inv) & M29;
        acc += (u64)m[k] * c_p29.p[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (u64)a.v[i] * b.v[k - i];
            acc += (u64)m[i] * c_p29.p[k - i];
        }
        r.v[k - 9] = (u32)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (u32)acc;
    return r;
}

// ---------------- variant C: 8 x 32-bit limbs, Comba with a 96-bit accumulator, carry via the MAD's carry-out
__device__ __forceinline__ void mac3(u64& acc, u32& ext, u32 a, u32 b) {
    u64 carry;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32_e64 %2, %1, 0, %2, %1"
        : "+v"(acc), "=&s"(carry), "+v"(ext)
        : "v"(a), "v"(b));
}
template <int MODE>
__device__ __forceinline__ void mac3m(u64& acc, u32& ext, u32 a, u32 b) {
    if (MODE == 2) {          // two statements: the scheduler may separate the MAD from the carry add
        u64 carry;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry) : "v"(a), "v"(b));
        asm("v_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(ext), "+s"(carry));
    } else {                  // carry through VCC
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ext) : "v"(a), "v"(b) : "vcc");
    }
}
template <class PP, int MODE>
__device__ __forceinline__ Fe<PP> mul_comba32m(const Fe<PP>& a, const Fe<PP>& b) {
    constexpr int N = PP::N;
    u32 m[N];
    Fe<PP> r;
    u64 acc = 0;
    u32 ext = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
            mac3m<MODE>(acc, ext, a.v[i], b.v[k - i]);
            mac3m<MODE>(acc, ext, m[i], PP::mod(k - i));
        }
        mac3m<MODE>(acc, ext, a.v[k], b.v[0]);
        m[k] = (u32)acc * PP::INV;
        mac3m<MODE>(acc, ext, m[k], PP::mod(0));
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            mac3m<MODE>(acc, ext, a.v[i], b.v[k - i]);
            mac3m<MODE>(acc, ext, m[i], PP::mod(k - i));
        }
        r.v[k - N] = (u32)acc;
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
    r.v[N - 1] = (u32)acc;
    fe_reduce_once(r);
    return r;
}
template <class PP>
__device__ __forceinline__ Fe<PP> mul_comba32(const Fe<PP>& a, const Fe<PP>& b) {
    constexpr int N = PP::N;
    u32 m[N];
    Fe<PP> r;
    u64 acc = 0;
    u32 ext = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
            mac3(acc, ext, a.v[i], b.v[k - i]);
            mac3(acc, ext, m[i], PP::mod(k - i));
        }
        mac3(acc, ext, a.v[k], b.v[0]);
        m[k] = (u32)acc * PP::INV;
        mac3(acc, ext, m[k], PP::mod(0));
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            mac3(acc, ext, a.v[i], b.v[k - i]);
            mac3(acc, ext, m[i], PP::mod(k - i));
        }
        r.v[k - N] = (u32)acc;
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
    r.v[N - 1] = (u32)acc;
    fe_reduce_once(r);
    return r;
}

// ---------------- variant D (rate probe only, not a multiplier): the FP64 inner pattern of a 5 x 52-bit
// limb product: per limb pair 2 FMAs (high and low half) and 2 integer 64-bit accumulations
__device__ __forceinline__ void fp_pair(double a, double b, long long& hi_acc, long long& lo_acc) {
    const double C1 = 0x1.0p104 + 0x1.0p103;   // forces the product's high half onto a fixed exponent
    double hi = __builtin_fma(a, b, C1);
    double lo = __builtin_fma(a, b, C1 - hi);
    hi_acc += __double_as_longlong(hi);
    lo_acc += __double_as_longlong(lo);
}

template <int V>
__global__ void k_bench(Fq* out, const Fq* in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = in[i], y = in[i + 1];
    if (V == 0) {
        for (int k = 0; k < iters; ++k) { x = fe_mul(x, y); y = fe_mul(y, x); }
    } else if (V == 2) {
        for (int k = 0; k < iters; ++k) { x = mul_comba32(x, y); y = mul_comba32(y, x); }
    } else if (V == 3) {
        for (int k = 0; k < iters; ++k) { x = mul_comba32m<P, 2>(x, y); y = mul_comba32m<P, 2>(y, x); }
    } else if (V == 4) {
        for (int k = 0; k < iters; ++k) { x = mul_comba32m<P, 3>(x, y); y = mul_comba32m<P, 3>(y, x); }
    }
    out[i] = fe_add(x, y);
}
__global__ void k_bench29(F29* out, const F29* in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x = in[i], y = in[i + 1];
    for (int k = 0; k < iters; ++k) { x = mul29(x, y); y = mul29(y, x); }
    F29 r;
    for (int q = 0; q < 9; ++q) r.v[q] = x.v[q] + y.v[q];
    out[i] = r;
}
__global__ void k_bench_fp(double* out, const double* in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double a[5], b[5];
    for (int q = 0; q < 5; ++q) { a[q] = in[i * 5 + q]; b[q] = in[i * 5 + q + 3]; }
    long long hi[10], lo[10];
    for (int q = 0; q < 10; ++q) hi[q] = lo[q] = 0;
    for (int k = 0; k < iters; ++k) {
        // two "multiplications" worth of limb products (product + reduction): 2 x 25 pairs
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int p = 0; p < 5; ++p)
#pragma unroll
                for (int q = 0; q < 5; ++q) fp_pair(a[p], b[q], hi[p + q], lo[p + q]);
        for (int q = 0; q < 5; ++q) { a[q] = (double)(hi[q] & 0xfffffffffffffll); b[q] = (double)(lo[q + 4] & 0xfffffffffffffll); }
    }
    double s = 0;
    for (int q = 0; q < 10; ++q) s += (double)(hi[q] ^ lo[q]);
    out[i] = s;
}
// one step of each exact variant, for the correctness dump
__global__ void k_once(Fq* o0, Fq* o2, const Fq* in, F29* o29, const F29* in29) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    o0[i] = fe_mul(in[i], in[i + 1]);
    o2[i] = mul_comba32(in[i], in[i + 1]);
    o29[i] = mul29(in29[i], in29[i + 1]);
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
static void to29(const u32* w, u32* l) {   // 256-bit little-endian words -> 9 x 29-bit limbs
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)w[wi] | (wi + 1 < 8 ? (u64)w[wi + 1] << 32 : 0);
        l[i] = (u32)(two >> sh) & M29;
    }
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    const size_t lanes = (size_t)blocks * threads;
    P29 hp;
    {
        u32 pw[8]; for (int i = 0; i < 8; ++i) pw[i] = P::mod(i);
        to29(pw, hp.p);
        u32 inv = 1;   // Newton: inv = p0^-1 mod 2^32
        for (int i = 0; i < 6; ++i) inv *= 2 - hp.p[0] * inv;
        hp.inv = (0u - inv) & M29;
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p29), &hp, sizeof(hp)));
    std::vector<u32> h((lanes + 2) * 8);
    for (auto& v : h) v = (u32)rand() * 2654435761u + (u32)rand();
    for (size_t i = 7; i < h.size(); i += 8) h[i] &= 0x1fffffff;
    std::vector<u32> h29((lanes + 2) * 9);
    for (size_t i = 0; i < lanes + 2; ++i) to29(&h[i * 8], &h29[i * 9]);
    Fq *in, *o0, *o2; F29 *in29, *o29;
    CK(hipMalloc(&in, (lanes + 2) * sizeof(Fq))); CK(hipMalloc(&o0, lanes * sizeof(Fq))); CK(hipMalloc(&o2, lanes * sizeof(Fq)));
    CK(hipMalloc(&in29, (lanes + 2) * sizeof(F29))); CK(hipMalloc(&o29, lanes * sizeof(F29)));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(in29, h29.data(), h29.size() * 4, hipMemcpyHostToDevice));
    double* fin; double* fout;
    CK(hipMalloc(&fin, (lanes * 5 + 8) * sizeof(double))); CK(hipMalloc(&fout, lanes * sizeof(double)));
    {
        std::vector<double> hd(lanes * 5 + 8);
        for (auto& d : hd) d = (double)(((u64)rand() << 21) ^ rand());
        CK(hipMemcpy(fin, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const int it = 512;
    auto report = [&](const char* name, float ms) {
        printf("%-34s %8.3f ms  %8.2f Gmul/s\n", name, ms, lanes * 2.0 * it / ms * 1e-6);
    };
    report("A  CIOS 8x32 (compiler)", time_it([&] { hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(threads), 0, 0, o0, in, it); }));
    report("B  Comba 9x29 lazy (compiler)", time_it([&] { hipLaunchKernelGGL(k_bench29, dim3(blocks), dim3(threads), 0, 0, o29, in29, it); }));
    report("C  Comba 8x32 + asm carry", time_it([&] { hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(threads), 0, 0, o2, in, it); }));
    report("C2 Comba 8x32, split asm", time_it([&] { hipLaunchKernelGGL(k_bench<3>, dim3(blocks), dim3(threads), 0, 0, o2, in, it); }));
    report("C3 Comba 8x32, carry in VCC", time_it([&] { hipLaunchKernelGGL(k_bench<4>, dim3(blocks), dim3(threads), 0, 0, o2, in, it); }));
    {   // correctness of C2 / C3 against A on the benchmark's own output
        std::vector<u32> ra(256 * 8), rc(256 * 8);
        hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(256), 0, 0, o0, in, 3); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ra.data(), o0, ra.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 3; v <= 4; ++v) {
            if (v == 3) hipLaunchKernelGGL(k_bench<3>, dim3(1), dim3(256), 0, 0, o2, in, 3);
            else hipLaunchKernelGGL(k_bench<4>, dim3(1), dim3(256), 0, 0, o2, in, 3);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(rc.data(), o2, rc.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0; for (size_t i = 0; i < ra.size(); ++i) bad += ra[i] != rc[i];
            printf("C%d vs A mismatching words: %d\n", v - 1, bad);
        }
    }
    report("D  FP64 pattern 5x52 (rate only)", time_it([&] { hipLaunchKernelGGL(k_bench_fp, dim3(blocks), dim3(threads), 0, 0, fout, fin, it); }));
    // correctness dump
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, o0, o2, in, o29, in29);
    CK(hipDeviceSynchronize());
    std::vector<u32> r0(64 * 8), r2(64 * 8), r29(64 * 9);
    CK(hipMemcpy(r0.data(), o0, r0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r29.data(), o29, r29.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r2[i];
    printf("C vs A mismatching words: %d\n", bad);
    for (int i = 0; i < 4; ++i) {
        printf("CHECK a=");
        for (int q = 7; q >= 0; --q) printf("%08x", h[i * 8 + q]);
        printf(" b=");
        for (int q = 7; q >= 0; --q) printf("%08x", h[(i + 1) * 8 + q]);
        printf(" A=");
        for (int q = 7; q >= 0; --q) printf("%08x", r0[i * 8 + q]);
        printf(" B29=");
        for (int q = 0; q < 9; ++q) printf("%x%s", r29[i * 9 + q], q < 8 ? "," : "");
        printf("\n");
    }
    return 0;
}

// tools/accum_alt_bench.hip — the two accumulation designs the product does NOT use, measured on the product's own field and
// curve arithmetic and table shape (development probe; VERDICT r4 item 3: "put them on the scoreboard, then close them"):
//
//   (i)  workgroup-level BATCHED-AFFINE accumulation: every lane keeps its running sum in AFFINE coordinates and has ONE pending
//        addition per step; the 256 denominators x2 - x1 of a workgroup are inverted together — product tree through LDS, ONE field
//        inversion per workgroup and step, back-substitution down the tree — and each lane finishes its affine addition with
//        3 products (lambda = dy / dx, lambda^2, lambda (x1 - x3)).  Fewer products per addition than XYZZ's 8M + 2S, IF the
//        inversion is shared by enough additions.
//  (ii)  LDS-staged WAVEFRONT SEGMENTED REDUCTION (north_star's wording): the lanes of a wavefront combine the partial sums they
//        end their slices with among themselves (log2 64 = 6 rounds of general XYZZ additions through LDS) instead of storing one
//        partial per lane for the fold kernels.
//
// Both run the same synthetic work as tools/accum_bench.hip: 2^24 additions per table launch, entries drawn at random from a
// 16-level table of 2^20 packed points (one 64-byte line per gather), slices of equal length.  `plain` is the product's hot
// addition in the same simplified loop (no bucket boundaries): the figure the alternatives are held against.
// The probes compute throw-away sums: (i) ignores the exceptional cases (equal x) that random data never meets, (ii) adds partials
// of different buckets — the instruction streams are the real ones, the results are not used.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/accum_alt_bench.hip -o tools/accum_alt_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../zokrates_amd/csrc/kernels_msm.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef Fu<Bn254Fq> F;

// ---- plain: the product's hot mixed addition, one slice of `len` entries per lane ----
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_plain(const AffPacked<F>* __restrict__ bases, const u32* __restrict__ sorted, u32 len, Xyzz<F>* __restrict__ out) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    const u32* mine = sorted + (size_t)g * len;
    u32 w[16];
    aff_load_words<F>(bases, mine[0] & 0x7fffffffu, w);
    Aff<F> p0 = aff_unpack<F>(w);
    Xyzz<F> a{p0.x, p0.y, F::one(), F::one()};
    aff_load_words<F>(bases, mine[1] & 0x7fffffffu, w);
    for (u32 i = 1; i < len; ++i) {
        Aff<F> pt = aff_unpack<F>(w);
        ZK_PIN_WORDS(pt);
        const u32 e = mine[i + 1 < len ? i + 1 : i];
        aff_load_words<F>(bases, e & 0x7fffffffu, w);
        F Pp, R;
        xyzz_madd_begin<true>(a, pt.x, pt.y, Pp, R);
        xyzz_madd_finish<true>(a, Pp, R);
    }
    out[g] = a;
}

// ---- (ii): the same loop, then the wavefront's 64 partial sums combined in 6 rounds through LDS ----
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_wave_segred(const AffPacked<F>* __restrict__ bases, const u32* __restrict__ sorted, u32 len, Xyzz<F>* __restrict__ out) {
    __shared__ u32 ex[36 * 256];
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    const u32* mine = sorted + (size_t)g * len;
    u32 w[16];
    aff_load_words<F>(bases, mine[0] & 0x7fffffffu, w);
    Aff<F> p0 = aff_unpack<F>(w);
    Xyzz<F> a{p0.x, p0.y, F::one(), F::one()};
    aff_load_words<F>(bases, mine[1] & 0x7fffffffu, w);
    for (u32 i = 1; i < len; ++i) {
        Aff<F> pt = aff_unpack<F>(w);
        ZK_PIN_WORDS(pt);
        const u32 e = mine[i + 1 < len ? i + 1 : i];
        aff_load_words<F>(bases, e & 0x7fffffffu, w);
        F Pp, R;
        xyzz_madd_begin<true>(a, pt.x, pt.y, Pp, R);
        xyzz_madd_finish<true>(a, Pp, R);
    }
    // segmented reduction inside the wavefront: round d adds the partial of lane + d where it belongs to the same bucket (here:
    // every other pair, so that half the lanes add in every round as they would on a sorted list with buckets a few slices wide)
    const u32 lane = threadIdx.x & 63, wbase = threadIdx.x & ~63u;
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32* av = (const u32*)&a;
        ZK_UNROLL for (int q = 0; q < 36; ++q) ex[q * 256 + threadIdx.x] = av[q];
        __builtin_amdgcn_wave_barrier();
        Xyzz<F> o;
        u32* ov = (u32*)&o;
        const u32 src = wbase + ((lane + d) & 63);
        ZK_UNROLL for (int q = 0; q < 36; ++q) ov[q] = ex[q * 256 + src];
        __builtin_amdgcn_wave_barrier();
        if ((lane / d) % 2 == 0) xyzz_add_acc(a, o);
    }
    if (lane == 0) out[g >> 6] = a;
}

// ---- (i): batched-affine, one pending addition per lane, the workgroup shares the inversion ----
// LDS: tree[2 * 256] field elements (9 words each, word-major): leaves 256 .. 511 hold the denominators, node k the product of
// its children 2k and 2k + 1; back-substitution turns node k into the inverse of what it held.
__device__ __forceinline__ F lds_rd(const u32* t, u32 node) { F r; ZK_UNROLL for (int q = 0; q < 9; ++q) r.v[q] = t[q * 512 + node]; return r; }
__device__ __forceinline__ void lds_wr(u32* t, u32 node, const F& v) { ZK_UNROLL for (int q = 0; q < 9; ++q) t[q * 512 + node] = v.v[q]; }
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_batched_affine(const AffPacked<F>* __restrict__ bases, const u32* __restrict__ sorted, u32 len, Aff<F>* __restrict__ out) {
    __shared__ u32 tree[9 * 512];
    const u32 t = threadIdx.x, g = blockIdx.x * blockDim.x + t;
    const u32* mine = sorted + (size_t)g * len;
    u32 w[16];
    aff_load_words<F>(bases, mine[0] & 0x7fffffffu, w);
    Aff<F> s = aff_unpack<F>(w);
    aff_load_words<F>(bases, mine[1] & 0x7fffffffu, w);
    for (u32 i = 1; i < len; ++i) {
        Aff<F> pt = aff_unpack<F>(w);
        ZK_PIN_WORDS(pt);
        const u32 e = mine[i + 1 < len ? i + 1 : i];
        aff_load_words<F>(bases, e & 0x7fffffffu, w);
        const F dx = fe_sub_k<4>(pt.x, s.x), dy = fe_sub_k<4>(pt.y, s.y);
        lds_wr(tree, 256 + t, dx);
        __syncthreads();
        for (u32 n = 128; n >= 1; n >>= 1) {                 // products up the tree: 8 levels
            if (t < n) lds_wr(tree, n + t, fu_mul_inl(lds_rd(tree, 2 * (n + t)), lds_rd(tree, 2 * (n + t) + 1)));
            __syncthreads();
        }
        if (t < 64) {                                        // ONE inversion per workgroup and step (every lane of the first wavefront
            const F inv = fu_inv(lds_rd(tree, 1));           // computes the same one: a wavefront costs the same with 1 or 64 lanes)
            if (t == 0) lds_wr(tree, 1, inv);
        }
        __syncthreads();
        for (u32 n = 1; n <= 128; n <<= 1) {                 // inverses down the tree: inv(left) = inv(parent) * right, and vice versa
            if (t < n) {
                const F ip = lds_rd(tree, n + t), l = lds_rd(tree, 2 * (n + t)), r = lds_rd(tree, 2 * (n + t) + 1);
                lds_wr(tree, 2 * (n + t), fu_mul_inl(ip, r));
                lds_wr(tree, 2 * (n + t) + 1, fu_mul_inl(ip, l));
            }
            __syncthreads();
        }
        const F lam = fu_mul_inl(dy, lds_rd(tree, 256 + t));
        const F x3 = fe_relax(fe_sub_k<4>(fe_sub_k<4>(fu_sqr_inl(lam), s.x), pt.x));
        s.y = fe_relax(fe_sub_k<4>(fu_mul_inl(lam, fe_sub_k<4>(s.x, x3)), s.y));
        s.x = x3;
        __syncthreads();
    }
    out[g] = s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const u32 npts = 1u << 20, levels = 16;
    const u64 total = (u64)1 << 24;
    std::vector<u32> hb((size_t)npts * levels * 16);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (i % 8 == 7) ? ((u32)rand() & 0x0fffffffu) : ((u32)rand() * 2654435761u);
    std::vector<u32> hs(total);
    for (auto& v : hs) v = (u32)(((((u64)rand() << 16) ^ (u64)rand()) % ((u64)npts * levels)));
    AffPacked<F>* bases; u32* sorted; void* out;
    CK(hipMalloc(&bases, hb.size() * 4)); CK(hipMemcpy(bases, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sorted, hs.size() * 4)); CK(hipMemcpy(sorted, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)cus * 4 * 64 * 8 * sizeof(Xyzz<F>)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        return best;
    };
    for (int waves : {4, 6, 8}) {
        const u32 nlanes = (u32)cus * 4 * 64 * waves, len = (u32)(total / nlanes);
        const double adds = (double)nlanes * (len - 1);
        float ms = time([&] { hipLaunchKernelGGL((k_plain<4>), dim3(nlanes / 256), dim3(256), 0, 0, bases, sorted, len, (Xyzz<F>*)out); });
        printf("plain XYZZ mixed addition (the product's)   slices/lane=%d len=%4u | %8.3f ms | %6.2f G additions/s\n", waves, len, ms, adds / ms * 1e-6);
        ms = time([&] { hipLaunchKernelGGL((k_wave_segred<4>), dim3(nlanes / 256), dim3(256), 0, 0, bases, sorted, len, (Xyzz<F>*)out); });
        printf("(ii) + wavefront segmented reduction (6 rounds) slices/lane=%d len=%4u | %8.3f ms | %6.2f G additions/s\n", waves, len, ms, adds / ms * 1e-6);
    }
    // (i) is two orders of magnitude slower: a sixteenth of the work is enough to time it
    for (int waves : {4, 8}) {
        const u32 nlanes = (u32)cus * 4 * 64 * waves, len = 17;
        const double adds = (double)nlanes * (len - 1);
        const float ms = time([&] { hipLaunchKernelGGL((k_batched_affine<4>), dim3(nlanes / 256), dim3(256), 0, 0, bases, sorted, len, (Aff<F>*)out); });
        printf("(i) batched-affine, one pending addition per lane, inversion shared by 256 lanes  slices/lane=%d len=%4u | %8.3f ms | %6.2f G additions/s\n", waves, len, ms, adds / ms * 1e-6);
    }
    return 0;
}

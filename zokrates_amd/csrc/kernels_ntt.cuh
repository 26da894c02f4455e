// kernels_ntt.cuh — element-wise field kernels, sparse R1CS mat-vec (K1) and the radix-2 NTT passes (K2-K4).
//
// Device replacement for [UPSTREAM] `LibsnarkReduction::witness_map` + `Radix2EvaluationDomain`
// reached from /root/reference/zokrates_ark/src/groth16.rs:44.
//
// NTT structure (four-step, N = N1*N2, both <= 2^11): a transform is two passes over HBM,
//   "cols" pass: N2/C workgroups, each stages C adjacent columns x N1 rows in LDS (C*32 B contiguous
//                per row -> full 128-B lines for C = 4), runs the N1-point sub-NTT there,
//   "rows" pass: each workgroup stages R contiguous rows of N2 elements and runs the N2-point sub-NTT.
// Every element-wise factor (inter-pass twiddle w^(a*b), 1/N, coset powers g^i, the Montgomery exit)
// is a table multiply fused into a pass's load ("pre") or store ("post"), so a transform touches each
// element exactly twice.  Natural-order input gives the digit-swapped "sigma" order
//   position p = k1*N2 + k2  holds  X[k1 + N1*k2]
// and sigma-order input gives natural-order output; the prover alternates the two kinds and never
// permutes (h_query is permuted once at key load instead).
#pragma once
#include "devrt.h"
#include "field.cuh"

namespace zk {

// ---------------- element-wise ----------------
template <class F>
__global__ void k_to_mont(const F* __restrict__ in, F* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_to_mont(in[i]);
}
template <class F>
__global__ void k_from_mont(const F* __restrict__ in, F* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_from_mont(in[i]);
}
// op: 0 add, 1 sub, 2 mul — operands and result in Montgomery form
template <class F>
__global__ void k_field_op(const F* __restrict__ a, const F* __restrict__ b, F* __restrict__ out, u64 n, int op) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = a[i], y = b[i];
    out[i] = op == 0 ? fe_add(x, y) : op == 1 ? fe_sub(x, y) : fe_mul(x, y);
}
// out[i] = in[i] * tbl[i]
template <class F>
__global__ void k_mul_table(const F* __restrict__ in, const F* __restrict__ tbl, F* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul(in[i], tbl[i]);
}
// out[i] = scale * base^e(i); mode 0: e = i; mode 1: e = (i / n2) * (i % n2)  (2-D twiddle);
// mode 2: e = natural index of sigma position i, i.e. k1 + n1*k2 with k1 = i / n2, k2 = i % n2
template <class F>
__global__ void k_pow_table(F* __restrict__ out, F base, F scale, u64 n, u32 n1, u32 n2, int mode) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 e = i;
    if (mode == 1) e = (i / n2) * (i % n2);
    else if (mode == 2) e = (i / n2) + (u64)n1 * (i % n2);
    out[i] = fe_mul(scale, fe_pow_u64(base, e));
}
// out[p] = in[sigma(p)] : gather between natural and sigma order (sigma is its own inverse only when n1 == n2,
// so both directions are provided): dir 0: out[k1*n2+k2] = in[k1 + n1*k2]; dir 1: out[k1 + n1*k2] = in[k1*n2+k2]
template <class F>
__global__ void k_sigma_permute(const F* __restrict__ in, F* __restrict__ out, u64 n, u32 n1, u32 n2, int dir) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    u64 nat = (p / n2) + (u64)n1 * (p % n2);
    if (dir == 0) out[p] = in[nat];
    else out[nat] = in[p];
}
// quotient evaluations: out = (a*b - c) * zinv        (App. A.3: division by the constant Z(g) on the coset)
template <class F>
__global__ void k_quotient(const F* __restrict__ a, const F* __restrict__ b, const F* __restrict__ c, F zinv, F* __restrict__ out,
                           u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul(fe_sub(fe_mul(a[i], b[i]), c[i]), zinv);
}

// ---------------- K1: sparse mat-vec ----------------
struct CsrDev {
    const u64* rowptr;
    const u32* col;
    const void* val;   // F[nnz], Montgomery
};
// grid.y = 3 (A, B, C).  out_m[i] = <M_i, z> for i < n; A additionally gets z[j] at n + j (j < l)
// ("input consistency" rows); everything up to N is zero-filled.
// A row is shared by G consecutive work-items (G = gA/gB/gC, a power of two <= 64 chosen from the matrix's average row
// length): compiler-made circuits have combinations of tens to hundreds of terms (every partial Poseidon round, every
// SHA-256 word sum), which one work-item per row would walk serially with a dependent random gather per term.  The G
// partial sums meet in an LDS tree.  G = 1 is the plain one-row-per-work-item kernel.
template <class F>
__global__ void __launch_bounds__(256) k_matvec(CsrDev A, CsrDev B, CsrDev C, const F* __restrict__ z, F* __restrict__ oa, F* __restrict__ ob,
                                                F* __restrict__ oc, u64 n, u64 l, u64 N, int gA, int gB, int gC) {
    __shared__ F sh[256];
    const int which = blockIdx.y;
    const int G = which == 0 ? gA : which == 1 ? gB : gC;
    const u32 rows_per_block = blockDim.x / (u32)G;
    if ((u64)blockIdx.x * rows_per_block >= N) return;            // whole workgroup past the end (uniform)
    const u64 i = (u64)blockIdx.x * rows_per_block + threadIdx.x / (u32)G;
    const u32 lane = threadIdx.x % (u32)G;
    const CsrDev M = which == 0 ? A : which == 1 ? B : C;
    F* out = which == 0 ? oa : which == 1 ? ob : oc;
    F acc = F::zero();
    if (i < n) {
        const F* val = (const F*)M.val;
        const u64 e = M.rowptr[i + 1];
        for (u64 k = M.rowptr[i] + lane; k < e; k += (u32)G) acc = fe_add(acc, fe_mul(val[k], z[M.col[k]]));
    } else if (which == 0 && i < n + l && lane == 0) {
        acc = z[i - n];
    }
    if (G > 1) {
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (u32 st = (u32)G >> 1; st > 0; st >>= 1) {
            if (lane < st) {
                acc = fe_add(acc, sh[threadIdx.x + st]);
                sh[threadIdx.x] = acc;
            }
            __syncthreads();
        }
    }
    if (i < N && lane == 0) out[i] = acc;
}
// lanes per row for a matrix with `nnz` entries in `n` rows: the largest power of two <= half the average row length
static inline int matvec_group(u64 nnz, u64 n) {
    const u64 avg = n ? nnz / n : 0;
    int g = 1;
    while (g < 64 && (u64)g * 4 <= avg) g *= 2;
    return g;
}

// ---------------- LDS-resident sub-NTT ----------------
// Elements live in two uint4 planes (limbs 0-3 / 4-7) so that consecutive lanes touch consecutive
// 16-B slots (conflict-free ds_read_b128/ds_write_b128); sequences are padded by one slot.
template <class F>
__device__ __forceinline__ F lds_get(const uint4* lo, const uint4* hi, int idx) {
    static_assert(F::N == 8, "Fr is 8 x 32-bit limbs");
    uint4 a = lo[idx], b = hi[idx];
    F r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <class F>
__device__ __forceinline__ void lds_put(uint4* lo, uint4* hi, int idx, const F& r) {
    lo[idx] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    hi[idx] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// In-place radix-2 DIF over `nseq` sequences of n = 2^logn points (stride n+1 slots); result is left in
// bit-reversed index order.  roots[j * rstride] = w_n^j for j < n/2.
template <class F>
__device__ __forceinline__ void lds_ntt_dif(uint4* lo, uint4* hi, int logn, int nseq, const F* __restrict__ roots, int rstride) {
    const int n = 1 << logn;
    const int halfn = n >> 1;
    const int total = nseq * halfn;
    const int nthreads = blockDim.x;
    for (int s = 0; s < logn; ++s) {
        const int half = halfn >> s;
        for (int b = threadIdx.x; b < total; b += nthreads) {
            const int seq = b / halfn, bb = b - seq * halfn;
            const int grp = bb / half, pos = bb - grp * half;
            const int i0 = seq * (n + 1) + grp * 2 * half + pos;
            const int i1 = i0 + half;
            F u = lds_get<F>(lo, hi, i0), v = lds_get<F>(lo, hi, i1);
            F d = fe_sub(u, v);
            if (pos) d = fe_mul(d, roots[(size_t)(pos << s) * rstride]);
            lds_put(lo, hi, i0, fe_add(u, v));
            lds_put(lo, hi, i1, d);
        }
        __syncthreads();
    }
}
static __device__ __forceinline__ int bitrev_n(int x, int logn) { return logn ? (int)(__brev((unsigned)x) >> (32 - logn)) : 0; }

// "cols" pass: the matrix is n1 x n2 row-major; this workgroup owns columns [c0, c0 + C).
template <class F>
__global__ void k_ntt_cols(F* __restrict__ data, int log_n1, u32 n2, int C, const F* __restrict__ roots, int rstride,
                           const F* __restrict__ pre, const F* __restrict__ post) {
    ZK_DYN_SMEM(smem);
    const int n1 = 1 << log_n1;
    uint4* lo = (uint4*)smem;
    uint4* hi = lo + (size_t)C * (n1 + 1);
    const u32 c0 = blockIdx.x * C;
    const int total = C * n1;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int a = e / C, j = e - a * C;
        const size_t g = (size_t)a * n2 + c0 + j;
        F x = data[g];
        if (pre) x = fe_mul(x, pre[g]);
        lds_put(lo, hi, j * (n1 + 1) + a, x);
    }
    __syncthreads();
    lds_ntt_dif<F>(lo, hi, log_n1, C, roots, rstride);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int k = e / C, j = e - k * C;
        const size_t g = (size_t)k * n2 + c0 + j;
        F x = lds_get<F>(lo, hi, j * (n1 + 1) + bitrev_n(k, log_n1));
        if (post) x = fe_mul(x, post[g]);
        data[g] = x;
    }
}

// "rows" pass: this workgroup owns rows [r0, r0 + R) of n2 contiguous elements each.
template <class F>
__global__ void k_ntt_rows(F* __restrict__ data, int log_n2, int R, const F* __restrict__ roots, int rstride,
                           const F* __restrict__ pre, const F* __restrict__ post) {
    ZK_DYN_SMEM(smem);
    const int n2 = 1 << log_n2;
    uint4* lo = (uint4*)smem;
    uint4* hi = lo + (size_t)R * (n2 + 1);
    const size_t base = (size_t)blockIdx.x * R * n2;
    const int total = R * n2;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = e >> log_n2, i = e & (n2 - 1);
        F x = data[base + e];
        if (pre) x = fe_mul(x, pre[base + e]);
        lds_put(lo, hi, r * (n2 + 1) + i, x);
    }
    __syncthreads();
    lds_ntt_dif<F>(lo, hi, log_n2, R, roots, rstride);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = e >> log_n2, k = e & (n2 - 1);
        F x = lds_get<F>(lo, hi, r * (n2 + 1) + bitrev_n(k, log_n2));
        if (post) x = fe_mul(x, post[base + e]);
        data[base + e] = x;
    }
}

}  // namespace zk

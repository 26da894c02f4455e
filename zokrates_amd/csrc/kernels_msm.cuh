// kernels_msm.cuh — Pippenger multi-scalar multiplication on the device (K5-K8).
//
// Device replacement for [UPSTREAM] `VariableBaseMSM::multi_scalar_mul` (ark-ec 0.3.0, SURVEY.md
// App. A.5), the five calls of which dominate `Groth16::prove`
// (/root/reference/zokrates_ark/src/groth16.rs:44).  Same mathematics (bucket method), different
// schedule: instead of one CPU thread per window walking all scalars, the device
//   1. recodes every scalar into W signed c-bit digits (half the buckets: 2^(c-1) per window),
//   2. counting-sorts the (window, bucket) keys so that every bucket's points are contiguous,
//   3. accumulates each bucket with XYZZ mixed additions (one work-item per bucket, bases gathered
//      as whole 64-B / 128-B affine points),
//   4. folds each window's buckets with the running-sum trick, many work-items per window plus an
//      LDS tree, leaving W window sums for the host's Horner step.
// One digit/sort pass is shared by every base set that uses the same scalars (a_query, b_g1_query,
// b_g2_query and l_query all pair with z).  The result is the exact group element, so it is
// independent of c, of the bucket order and of how ark itself schedules the sum.
#pragma once
#include "devrt.h"
#include "ec.cuh"

namespace zk {

static constexpr u32 MSM_NO_DIGIT = 0xffffffffu;

// ---- 1. signed-digit recoding + bucket histogram ----
// scalars: n x 8 u32 canonical.  dig[j*n + i] = bucket | sign<<31 (bucket = |d|-1) or MSM_NO_DIGIT.
// cnt[j*K + bucket] += 1.
static __global__ void k_msm_digits(const u32* __restrict__ scalars, u64 n, int c, int W, u32* __restrict__ dig, u32* __restrict__ cnt) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[9];
    const uint4* sp = (const uint4*)(scalars + i * 8);
    uint4 lo = sp[0], hi = sp[1];
    k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w;
    k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
    k[8] = 0;
    const u32 K = 1u << (c - 1);
    const u32 mask = (1u << c) - 1;
    u32 carry = 0;
    for (int j = 0; j < W; ++j) {
        const int bit = j * c, limb = bit >> 5, off = bit & 31;
        u32 raw = 0;
        if (limb < 8) {
            u64 two = (u64)k[limb] | ((u64)k[limb + 1] << 32);
            raw = (u32)(two >> off) & mask;
        }
        raw += carry;
        u32 out = MSM_NO_DIGIT;
        if (raw > K) {            // digit = raw - 2^c (negative), borrow one from the next window
            u32 mag = (1u << c) - raw;
            out = (mag - 1) | 0x80000000u;
            carry = 1;
        } else {
            carry = 0;
            if (raw) out = raw - 1;
        }
        dig[(u64)j * n + i] = out;
        if (out != MSM_NO_DIGIT) atomicAdd(&cnt[(u64)j * K + (out & 0x7fffffffu)], 1u);
    }
}

// ---- 2a. exclusive scan of the W*K counters: one workgroup per chunk of SCAN_CHUNK counters ----
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_PER_THREAD = 16;
static constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;

static __global__ void k_scan_local(const u32* __restrict__ cnt, u32* __restrict__ off, u32* __restrict__ chunk_sum, u64 total) {
    __shared__ u32 sh[SCAN_THREADS];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_PER_THREAD;
    u32 v[SCAN_PER_THREAD];
    u32 s = 0;
    for (int q = 0; q < SCAN_PER_THREAD; ++q) {
        u32 x = base + q < total ? cnt[base + q] : 0;
        v[q] = s;
        s += x;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {   // Hillis-Steele inclusive scan of the per-thread sums
        u32 t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    const u32 excl = sh[threadIdx.x] - s;
    for (int q = 0; q < SCAN_PER_THREAD; ++q)
        if (base + q < total) off[base + q] = excl + v[q];
    if (threadIdx.x == SCAN_THREADS - 1) chunk_sum[blockIdx.x] = sh[threadIdx.x];
}
// single workgroup: exclusive scan of the chunk sums in place (nchunks <= a few thousand); writes the grand total
static __global__ void k_scan_chunks(u32* __restrict__ chunk_sum, u32 nchunks, u32* __restrict__ grand_total) {
    __shared__ u32 sh[SCAN_THREADS];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 b = 0; b < nchunks; b += SCAN_THREADS) {
        u32 i = b + threadIdx.x;
        u32 x = i < nchunks ? chunk_sum[i] : 0;
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int d = 1; d < SCAN_THREADS; d <<= 1) {
            u32 t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nchunks) chunk_sum[i] = carry + sh[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
static __global__ void k_scan_add(u32* __restrict__ off, const u32* __restrict__ chunk_sum, u64 total, const u32* __restrict__ grand_total) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) off[i] += chunk_sum[i / SCAN_CHUNK];
    if (i == total) off[total] = *grand_total;   // sentinel: off has total+1 entries
}

// ---- 2b. scatter point indices into bucket order ----
static __global__ void k_msm_scatter(const u32* __restrict__ dig, u64 n, int c, int W, const u32* __restrict__ off, u32* __restrict__ cursor,
                              u32* __restrict__ sorted) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (u64)W) return;
    const u32 d = dig[t];
    if (d == MSM_NO_DIGIT) return;
    const u64 j = t / n, i = t - j * n;
    const u32 K = 1u << (c - 1);
    const u64 key = j * K + (d & 0x7fffffffu);
    const u32 pos = atomicAdd(&cursor[key], 1u);
    sorted[off[key] + pos] = (u32)i | (d & 0x80000000u);
}

// ---- 3. bucket accumulation: one work-item per (window, bucket) ----
template <class F>
__global__ void k_msm_accum(const Aff<F>* __restrict__ bases, const u32* __restrict__ off, const u32* __restrict__ sorted,
                            Xyzz<F>* __restrict__ buckets, u64 nbuckets) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    const u32 beg = off[b], end = off[b + 1];
    Xyzz<F> acc = Xyzz<F>::inf();
    for (u32 k = beg; k < end; ++k) {
        const u32 e = sorted[k];
        Aff<F> p = bases[e & 0x7fffffffu];
        if (e & 0x80000000u) p = aff_neg(p);
        acc = xyzz_madd(acc, p);
    }
    buckets[b] = acc;
}

// ---- 4. window fold: sum_{b} (b+1) * B_b ----
// Work-item t of window j owns buckets [t*L, (t+1)*L): running-sum over them gives
// sum (b - tL + 1) B_b and the plain sum S; the missing tL * S is a short double-and-add.  The
// workgroup then tree-adds its contributions in LDS and writes one partial per workgroup.
template <class F>
__global__ void k_msm_fold(const Xyzz<F>* __restrict__ buckets, u32 K, int L, Xyzz<F>* __restrict__ partial) {
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 j = blockIdx.y;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    Xyzz<F> contrib = Xyzz<F>::inf();
    if ((u64)t * L < K) {
        const Xyzz<F>* B = buckets + (u64)j * K + (u64)t * L;
        Xyzz<F> run = Xyzz<F>::inf(), acc = Xyzz<F>::inf();
        for (int q = L - 1; q >= 0; --q) {
            if ((u64)t * L + q < K) run = xyzz_add(run, B[q]);
            acc = xyzz_add(acc, run);
        }
        contrib = xyzz_add(acc, xyzz_mul_u32(run, t * (u32)L));
    }
    sh[threadIdx.x] = contrib;
    __syncthreads();
    for (unsigned s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(u64)j * gridDim.x + blockIdx.x] = sh[0];
}
// one workgroup per window: sum its `nparts` partials
template <class F>
__global__ void k_msm_fold_final(const Xyzz<F>* __restrict__ partial, u32 nparts, Xyzz<F>* __restrict__ window_sum) {
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 j = blockIdx.x;
    Xyzz<F> acc = Xyzz<F>::inf();
    for (u32 q = threadIdx.x; q < nparts; q += blockDim.x) acc = xyzz_add(acc, partial[(u64)j * nparts + q]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (unsigned s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) window_sum[j] = sh[0];
}

// ---- key preparation ----
// canonical affine coordinates (+ host-decoded infinity as all-zero) -> Montgomery form; for Fq2 points the
// coordinate array is simply twice as long, so this is an element-wise conversion over base-field elements.
// out[p] = (idx < n_src) ? in[idx] : infinity, idx = natural index of sigma position p (h_query layout)
template <class PT>
__global__ void k_sigma_gather_points(const PT* __restrict__ in, PT* __restrict__ out, u64 n, u64 n_src, u32 n1, u32 n2) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    u64 nat = (p / n2) + (u64)n1 * (p % n2);
    out[p] = nat < n_src ? in[nat] : PT::inf();
}

// fixed-base multiplication for setup (N3): out[i] = k_i * G with tbl[j*256 + d] = d * 2^(8j) * G (affine)
template <class F>
__global__ void k_fixed_base_mul(const u32* __restrict__ scalars, u64 n, const Aff<F>* __restrict__ tbl, int nwin,
                                 Aff<F>* __restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* k = scalars + i * 8;
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int j = 0; j < nwin; ++j) {
        const u32 d = (k[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        if (d) acc = xyzz_madd(acc, tbl[(size_t)j * 256 + d]);
    }
    out[i] = xyzz_to_affine(acc);
}

}  // namespace zk

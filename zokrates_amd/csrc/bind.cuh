// bind.cuh — binding a resident Groth16 proving key to ONE constraint system: the kernels.
//
// `Groth16::prove` (/root/reference/zokrates_ark/src/groth16.rs:44; [UPSTREAM] ark-groth16 0.3.0 `LibsnarkReduction::witness_map`,
// SURVEY.md App. A.3) turns the evaluations of a, b, c into the COEFFICIENTS of h = (ab - c)/Z — seven transforms in the reference,
// six in this library's default schedule (core.cuh witness_map) — only to pair them with h_query[i] = [tau^i Z(tau)/delta]_1.  The
// transforms are linear maps, so they can be applied to the BASES once instead of to the scalars in every proof:
//
//     sum_{i <= N-2} h_i H_i  =  sum_j U_j H'_j  +  sum_v z_v D_v ,          U_j = a(g w^j) b(g w^j) / Z(g)
//
//     H'_j = sum_{i <= N-2} (g^-i / N) w^(-ij) H_i            the coset inverse transform, transposed, on the bases
//     D_v  = sum_k C[k][v] H''_k,   H''_k = sum_{i <= N-2} (-1 / (N Z(g))) w^(-ik) H_i      c's inverse transform AND its mat-vec
//
// (the coefficient of x^(N-1) never meets a base — ark's h_query has N-1 entries — which is exactly "H_{N-1} = infinity" on the
// right-hand side).  D_v is added to l_query[v] (the padded l table: public variables get an entry too), H' replaces h_query.  A
// proof then takes FOUR transforms (a, b: to coefficients, to the coset) and the mat-vec of A and B only; its group elements
// A, B, C are the same, so the proof bytes are.  The price is paid once per (key, constraint system): two size-N transforms over
// G1 points (N/2 log2 N scalar multiplications each) — seconds at 2^20, for a prover that keeps its key resident.
//
// Kernels (G1 only, the unsaturated field of fieldu.cuh, XYZZ points): k_bind_scale (bases -> scaled XYZZ, natural order),
// k_bind_fft_stage (one radix-2 decimation-in-frequency stage over both vectors), k_bind_h_finish (bit-reversed XYZZ -> packed
// affine level 0 of the H' table), k_bind_cmul (one product C[k][v] H''_k per non-zero of C, in column order), k_bind_l_sum_short / _long / k_bind_l_affine
// (per variable: the sum of its products + l_query[v] -> packed affine level 0 of the L' table).
#pragma once
#include "kernels_msm.cuh"

namespace zk {

// (The kernels below carry the fold kernels' waves-per-SIMD target — MsmTuning::COLD_WPE: the out-of-line routines they share with
// them, xyzz_dbl first of all, are compiled once per translation unit for the LEAST demanding of their callers, and a caller without
// a target gave them the whole register file: the fold kernels then missed theirs.)
// -p, coordinates within the stored bounds again (Y < 3p)
template <class F>
ZK_HD Xyzz<F> xyzz_neg_u(const Xyzz<F>& p) {
    if (p.is_inf()) return p;
    return {p.x, fe_relax(fe_sub_k<4>(F::zero(), p.y)), p.zz, p.zzz};
}
// k * p for a canonical integer k of `nw` 32-bit words (read where they lie: `k` may point into HBM): two bits at a time against
// {p, 2p, 3p} (127 additions instead of the ~254 slots a wavefront pays for a bit-by-bit ladder, whose lanes disagree at every bit).
// The running point stays in REGISTERS: the doublings are inlined and in place (ec.cuh xyzz_dbl_acc), the additions fetch the table
// entry from the lane's scratch where they use it (xyzz_add_from); `p_mem`: the point, wherever the caller keeps it.  Round 5's form was a chain of
// out-of-line calls on points in scratch (xyzz_dbl, xyzz_add_acc_call: every operand through memory twice per call, a 2 KB frame):
// 42 ms per transform stage at 2^20 where the arithmetic alone is ~25.
template <class F>
ZK_HD Xyzz<F> xyzz_mul_words(const Xyzz<F>* p_mem, const u32* __restrict__ k, int nw) {
    Xyzz<F> r = Xyzz<F>::inf();
    if (p_mem->is_inf()) return r;
    Xyzz<F> tab[3];                                              // p, 2p, 3p
    {
        Xyzz<F> t = *p_mem;
        tab[0] = t;
        xyzz_dbl_acc(t);
        tab[1] = t;
        xyzz_add_from(t, &tab[0]);
        tab[2] = t;
    }
    int top = nw * 16 - 1;                                       // highest non-zero two-bit digit
    while (top >= 0 && ((k[top >> 4] >> ((top & 15) * 2)) & 3u) == 0) --top;
    for (int i = top; i >= 0; --i) {
        if (i != top) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int rep = 0; rep < 2; ++rep) xyzz_dbl_acc(r);
        }
        const u32 d = (k[i >> 4] >> ((i & 15) * 2)) & 3u;
        if (d) xyzz_add_from(r, &tab[d - 1]);
    }
    return r;
}
// (Round 6 tried signed four-bit windows against {p .. 8p} — 64 additions instead of ~95, ~20 % fewer field products on paper: the
// stage kernel took 48.9 ms instead of 41.8 (profiles/r6h_bound_serial_kernel_stats.md: a 2 864-byte frame, digits and table entries
// fetched through per-lane indices).  The two-bit form stays.)

// out[vec * n + nat] = s * H_nat in XYZZ, H_nat = tbl[p] with nat = sigma_nat(p) (level 0 of the key's h table: sigma order);
// natural indices >= n_src (the padding: N - 1) are the point at infinity.  vec 0: s = scal[nat] (g^-nat / N); vec 1: s = *konst.
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_scale(const AffPacked<F>* __restrict__ tbl, u64 n, u64 n_src, u32 n1, u32 n2, u32 n3,
                                                    const u32* __restrict__ scal, const u32* __restrict__ konst, int nw, Xyzz<F>* __restrict__ out) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const u64 nat = sigma_nat(p, n1, n2, n3);
    Xyzz<F> r = Xyzz<F>::inf();
    if (nat < n_src) {
        u32 w[2 * AffPacked<F>::NW];
        aff_load_words<F>(tbl, p, w);
        const Aff<F> a = aff_unpack<F>(w);
        // (the point itself is parked in its output slot: the first entry of the multiplication's table)
        Xyzz<F>* const slot = out + (u64)blockIdx.y * n + nat;
        *slot = Xyzz<F>::from_affine(a);
        r = xyzz_mul_words<F>(slot, blockIdx.y == 0 ? scal + nat * (u64)nw : konst, nw);
    }
    out[(u64)blockIdx.y * n + nat] = r;
}

// one radix-2 decimation-in-frequency stage, half-length q (sub-transform length 2q): for every pair (i, i + q)
//     x[i] <- x[i] + x[i+q],   x[i+q] <- tw[pos * (n / 2q)] * (x[i] - x[i+q]),   pos = i mod q;   tw[e] = w^-e as a canonical integer.
// blockIdx.y: which of the two vectors.  After log2 n stages position p holds the output of index bitrev(p).
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_fft_stage(Xyzz<F>* __restrict__ x, u64 n, u64 q, const u32* __restrict__ tw, int nw) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    Xyzz<F>* v = x + (u64)blockIdx.y * n;
    const u64 blk = t / q, pos = t - blk * q;
    const u64 i = blk * 2 * q + pos, j = i + q;
    // the difference first, held while the sum is made and stored; then it goes to its own slot, where the multiplication reads it
    Xyzz<F> d = v[i];
    xyzz_add_from(d, &v[j], true);
    {
        Xyzz<F> s = v[i];
        xyzz_add_from(s, &v[j]);
        v[i] = s;
    }
    v[j] = d;
    const u64 e = pos * (n / (2 * q));
    if (e != 0) v[j] = xyzz_mul_words<F>(&v[j], tw + e * (u64)nw, nw);
}

template <class F>
ZK_HD Aff<F> xyzz_to_affine_u(const Xyzz<F>& p) {
    if (p.is_inf()) return Aff<F>::inf();
    const F i3 = ec_inv(p.zzz);
    const F i2 = ec_sqr(ec_mul(p.zz, i3));
    return {ec_mul(p.x, i2), ec_mul(p.y, i3)};
}
static ZK_HD u64 bind_bitrev(u64 x, int logn) {
    u64 r = 0;
    for (int b = 0; b < logn; ++b) r |= ((x >> b) & 1) << (logn - 1 - b);
    return r;
}
// level 0 of the H' table: out[j] = affine(x[bitrev(j)]), packed
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_h_finish(const Xyzz<F>* __restrict__ x, u64 n, int logn, AffPacked<F>* __restrict__ out) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    aff_pack(xyzz_to_affine_u(x[bind_bitrev(j, logn)]), out + j);
}

// One product per non-zero of C, in COLUMN order (the host sorts C's entries by variable): prod[e] = val[e] * x[bitrev(row[e])].
// val: canonical integers; the coefficients 1 and p - 1, which are most of a compiled circuit, cost nothing.
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_cmul(const Xyzz<F>* __restrict__ x, int logn, const u32* __restrict__ row, const u32* __restrict__ val, int nw,
                                                   const u32* __restrict__ minus_one, u64 nnz, Xyzz<F>* __restrict__ prod) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const Xyzz<F>* const src = x + bind_bitrev(row[e], logn);
    bool is_one = true, is_m1 = true;
    for (int w = 0; w < nw; ++w) {
        const u32 kw = val[e * (u64)nw + w];
        is_one = is_one && kw == (w == 0 ? 1u : 0u);
        is_m1 = is_m1 && kw == minus_one[w];
    }
    if (is_one) prod[e] = *src;
    else if (is_m1) prod[e] = xyzz_neg_u(*src);
    else prod[e] = xyzz_mul_words<F>(src, val + e * (u64)nw, nw);
}

// The per-variable sums S_v = l[v] + sum of prod[cptr[v] .. cptr[v+1]) in three stages (round 5 ran ONE workgroup of 64 per variable
// whose lane 0 alone converted to affine form: 0.13 s at 2^20 with 63 lanes of 64 idle through a field inversion):
//   k_bind_l_sum_short  one work-item per variable: columns of at most BIND_SHORT_COL entries (all but a handful in a compiled circuit)
//   k_bind_l_sum_long   one workgroup per listed variable (the host lists the long columns: the constant ONE of a circuit full of
//                       `x * y == k` rows occurs in every row of C): work-items stride over the products and meet in an LDS tree
//   k_bind_l_affine     one work-item per variable: S_v -> packed affine (level 0 of the L' table), every lane inverting
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_l_sum_short(const Xyzz<F>* __restrict__ prod, const u64* __restrict__ cptr, const AffPacked<F>* __restrict__ l,
                                                         u64 m, Xyzz<F>* __restrict__ sum) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    const u64 b = cptr[v], e = cptr[v + 1];
    if (e - b > BIND_SHORT_COL) return;              // k_bind_l_sum_long writes it
    u32 w[2 * AffPacked<F>::NW];
    aff_load_words<F>(l, v, w);
    Xyzz<F> s = Xyzz<F>::from_affine(aff_unpack<F>(w));
    for (u64 i = b; i < e; ++i) xyzz_add_from(s, &prod[i]);
    sum[v] = s;
}
template <class F>
__global__ void __launch_bounds__(64, MsmTuning<F>::COLD_WPE) k_bind_l_sum_long(const Xyzz<F>* __restrict__ prod, const u64* __restrict__ cptr, const AffPacked<F>* __restrict__ l,
                                                        const u32* __restrict__ cols, Xyzz<F>* __restrict__ sum) {
    __shared__ Xyzz<F> sh[64];
    const u64 v = cols[blockIdx.x];
    const u64 b = cptr[v], e = cptr[v + 1];
    Xyzz<F> s = Xyzz<F>::inf();
    for (u64 i = b + threadIdx.x; i < e; i += 64) xyzz_add_from(s, &prod[i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (unsigned st = 32; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            xyzz_add_from(s, &sh[threadIdx.x + st]);
            sh[threadIdx.x] = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u32 w[2 * AffPacked<F>::NW];
        aff_load_words<F>(l, v, w);
        sh[1] = Xyzz<F>::from_affine(aff_unpack<F>(w));       // (slot 1 is free after the tree's last level)
        xyzz_add_from(s, &sh[1]);
        sum[v] = s;
    }
}
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_bind_l_affine(const Xyzz<F>* __restrict__ sum, u64 m, AffPacked<F>* __restrict__ out) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    aff_pack(xyzz_to_affine_u(sum[v]), out + v);
}

}  // namespace zk

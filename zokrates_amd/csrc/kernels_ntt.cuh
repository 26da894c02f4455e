// kernels_ntt.cuh — element-wise field kernels, sparse R1CS mat-vec (K1) and the radix-2 NTT passes (K2-K4).
//
// Device replacement for [UPSTREAM] `LibsnarkReduction::witness_map` + `Radix2EvaluationDomain`
// reached from /root/reference/zokrates_ark/src/groth16.rs:44.
//
// NTT structure (four-step, N = N1*N2, both <= 2^11): a transform is two passes over HBM,
//   "cols" pass: each workgroup stages C adjacent columns x N1 rows in LDS and runs the N1-point sub-NTTs there,
//   "rows" pass: each workgroup stages R contiguous rows of N2 elements and runs the N2-point sub-NTTs.
// Every element-wise factor (inter-pass twiddle w^(a*b), 1/N, coset powers g^i, the Montgomery exit)
// is a table multiply fused into a pass's store ("post"), so a transform touches each element exactly twice.
// Natural-order input gives the digit-swapped "sigma" order
//   position p = k1*N2 + k2  holds  X[k1 + N1*k2]
// and sigma-order input gives natural-order output; the prover alternates the two kinds and never
// permutes (h_query is permuted once at key load instead).
//
// Arithmetic: the passes compute in the UNSATURATED 9 x 29-bit representation of fieldu.cuh (lazy Comba Montgomery
// product with R' = 2^261: 1.9x the multiplier throughput of saturated 8 x 32-bit CIOS; both Fr moduli are <= 255 bits).
// The NTT vectors therefore hold x * R' mod p ("R'-form"), packed into the same 32 bytes per element (fu_pack); a value
// is unpacked when a pass loads it, lives as 9 limb planes in LDS, and is packed again when the pass stores it.
// Sub-NTTs are radix-4 decimation-in-frequency butterflies in registers (two radix-2 stages per LDS round trip and
// barrier; a last radix-2 stage when log2 n is odd): 4 field multiplications per butterfly, none in the last round.
// Kernels that touch R'-form data outside the passes (mat-vec output, quotient, GM17 rows) use the rp_* helpers below.
#pragma once
#include <vector>

#include "devrt.h"
#include "fieldu.cuh"

namespace zk {

// ---------------- element-wise ----------------
template <class F>
__global__ void k_to_mont(const F* __restrict__ in, F* __restrict__ out, u64 n) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_to_mont(in[i]);
}
template <class F>
__global__ void k_from_mont(const F* __restrict__ in, F* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_from_mont(in[i]);
}
// ---- R'-form helpers: canonical packed integers of x * R' mod p, held in Fe words (add / sub / dbl are the plain
// modular ones of field.cuh; only products need the R' Montgomery product) ----
template <class P>
ZK_HD Fe<P> rp_canon(const Fu<P>& r) {   // TIGHT, value < 2p -> canonical packed words
    Fe<P> o;
    fu_pack(r, o.v);
    fe_reduce_once(o);
    return o;
}
template <class P> ZK_HD Fe<P> rp_mul(const Fe<P>& a, const Fe<P>& b) { return rp_canon(fu_mul_inl(fu_unpack<P>(a.v), fu_unpack<P>(b.v))); }
template <class P> ZK_HD Fe<P> rp_sqr(const Fe<P>& a) { const Fu<P> u = fu_unpack<P>(a.v); return rp_canon(fu_sqr_inl(u)); }
template <class P> ZK_HD Fe<P> rp_one() { return rp_canon(Fu<P>::one()); }
template <class P>
ZK_HD Fe<P> rp_to_plain(const Fe<P>& a) {   // x * R' -> x (canonical integer)
    Fu<P> unit = Fu<P>::zero();
    unit.v[0] = 1;
    return rp_canon(fu_mul_inl(fu_unpack<P>(a.v), unit));
}
// out[i] = in[i] * k / R   (k = R' * R mod p turns a canonical integer into R'-form; k = R' mod p turns the saturated
// Montgomery form into R'-form)
template <class F>
__global__ void k_mul_const(const F* __restrict__ in, F* __restrict__ out, u64 n, F k) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul(in[i], k);
}
template <class F>
__global__ void k_from_rp(const F* __restrict__ in, F* __restrict__ out, u64 n) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rp_to_plain(in[i]);
}
// packed words -> the 9-limb working form (root tables of the sub-NTTs)
template <class P>
__global__ void k_unpack_table(const Fe<P>* __restrict__ in, Fu<P>* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fu_unpack<P>(in[i].v);
}
// flag |= 1 if any of the n scalars (canonical-integer words) is >= the modulus: such an entry would overflow the MSM's
// signed-digit recoding (ark's FromBytes rejects it)
template <class F>
__global__ void k_check_canonical(const F* __restrict__ x, u64 n, u32* __restrict__ flag) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F v = x[i];
    bool lt = false, decided = false;
    ZK_UNROLL for (int k = F::N - 1; k >= 0; --k) {
        if (!decided && v.v[k] != F::Params::mod(k)) { lt = v.v[k] < F::Params::mod(k); decided = true; }
    }
    if (!lt) atomicOr(flag, 1u);
}
// Position-keyed checksum of `nwords` 32-bit words: out[0], out[1] += two independent 64-bit mixes of (word, index, salt) summed over
// the array.  What a proving-key image remembers of the constraint system its bound tables were made for (zkhip_pk::bound_fp): a
// guard against pairing them with another system by accident, not a cryptographic commitment.
static __global__ void k_fingerprint(const u32* __restrict__ data, u64 nwords, u64 salt, unsigned long long* __restrict__ out) {
    u64 a = 0, b = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) {
        u64 x = ((u64)data[i] << 32 | (u32)i) ^ (salt + (i >> 32) * 0xD6E8FEB86659FD93ull);
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        a += x;
        b += (x * 0x9E3779B97F4A7C15ull) ^ (x >> 17);
    }
    if (a | b) {
        atomicAdd(&out[0], (unsigned long long)a);
        atomicAdd(&out[1], (unsigned long long)b);
    }
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
// out[i] = scale * base^e(i); mode 0: e = i; mode 1: e = (i / n2) * (i % n2)  (the twiddles between two passes: rows x n2
// columns); mode 2: e = natural index of sigma position i (sigma_nat, field.cuh)
template <class F>
__global__ void k_pow_table(F* __restrict__ out, F base, F scale, u64 n, u32 n1, u32 n2, int mode, u32 n3 = 1) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 e = i;
    if (mode == 1) e = (i / n2) * (i % n2);
    else if (mode == 2) e = sigma_nat(i, n1, n2, n3);
    out[i] = fe_mul(scale, fe_pow_u64(base, e));
}
// out[p] = in[sigma(p)] : gather between natural and sigma order (sigma is its own inverse only when n1 == n2,
// so both directions are provided): dir 0: out[p] = in[sigma_nat(p)]; dir 1: out[sigma_nat(p)] = in[p]
template <class F>
__global__ void k_sigma_permute(const F* __restrict__ in, F* __restrict__ out, u64 n, u32 n1, u32 n2, int dir, u32 n3 = 1) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    u64 nat = sigma_nat(p, n1, n2, n3);
    if (dir == 0) out[p] = in[nat];
    else out[nat] = in[p];
}
// quotient evaluations, product part: out = a*b * zinv   (App. A.3: division by the constant Z(g) on the coset); R'-form
// operands (any packed value < 2^256), zinv in R'-form, canonical R'-form out.  The `- c` of (a*b - c)/Z never visits the
// coset: the coset transform pair is the identity on it, so its share of the quotient's coefficients is zinv * c's own
// coefficients, subtracted where the last transform stores h (ntt_store's `minus`) — 6 transforms per proof, not 7.
template <class P>
__global__ void k_quotient(const Fe<P>* __restrict__ a, const Fe<P>* __restrict__ b, Fe<P> zinv, Fe<P>* __restrict__ out, u64 n) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fu<P> ab = fu_mul_inl(fu_unpack<P>(a[i].v), fu_unpack<P>(b[i].v));          // < 2p
    out[i] = rp_canon(fu_mul_inl(ab, fu_unpack<P>(zinv.v)));
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
// Rows of more than MATVEC_LONG terms are left to k_matvec_long (a wavefront per row): the reference's optimizer inlines every
// linear definition, so the sum check of a lazily added u32 of a SHA-256 round is ONE row over every wire the value was ever
// added up from — thousands of terms next to one-term boolean checks (zokrates_amd/sha256_circuit.py) — and G lanes walking
// such a row hold their whole workgroup for milliseconds.
static constexpr u32 MATVEC_LONG = 32;
template <class F>
__global__ void __launch_bounds__(256) k_matvec(CsrDev A, CsrDev B, CsrDev C, const F* __restrict__ z, F* __restrict__ oa, F* __restrict__ ob,
                                                F* __restrict__ oc, u64 n, u64 l, u64 N, int gA, int gB, int gC, u64 m_vars, int mat0 = 0) {
    ZK_PRIO_HIGH();
    __shared__ F sh[256];
    const int which = blockIdx.y + mat0;
    const int G = which == 0 ? gA : which == 1 ? gB : gC;
    const u32 rows_per_block = blockDim.x / (u32)G;
    if ((u64)blockIdx.x * rows_per_block >= N) return;            // whole workgroup past the end (uniform)
    const u64 i = (u64)blockIdx.x * rows_per_block + threadIdx.x / (u32)G;
    const u32 lane = threadIdx.x % (u32)G;
    const CsrDev M = which == 0 ? A : which == 1 ? B : C;
    F* out = which == 0 ? oa : which == 1 ? ob : oc;
    F acc = F::zero();
    bool mine = true;               // false: a long row, written by k_matvec_long
    if (i < n) {
        const F* val = (const F*)M.val;
        const u64 b = M.rowptr[i], e = M.rowptr[i + 1];
        mine = e - b <= MATVEC_LONG;
        if (mine)
            for (u64 k = b + lane; k < e; k += (u32)G) {
                ZK_ASSERT_IDX(M.col[k] < m_vars);
                acc = fe_add(acc, fe_mul(val[k], z[M.col[k]]));
            }
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
    if (i < N && lane == 0 && mine) out[i] = acc;
}
// one wavefront per long row (four rows per workgroup): rows[] = matrix << 32 | row, listed on the host when the system is loaded
template <class F>
__global__ void __launch_bounds__(256) k_matvec_long(CsrDev A, CsrDev B, CsrDev C, const F* __restrict__ z, F* __restrict__ oa, F* __restrict__ ob,
                                                     F* __restrict__ oc, const u64* __restrict__ rows, u64 n_long, u64 m_vars) {
    ZK_PRIO_HIGH();
    __shared__ F sh[256];
    const u32 lane = threadIdx.x & 63u;
    const u64 slot = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool live = slot < n_long;
    const u64 r = live ? rows[slot] : 0;
    const int which = (int)(r >> 32);
    const u64 i = r & 0xffffffffu;
    const CsrDev M = which == 0 ? A : which == 1 ? B : C;
    F acc = F::zero();
    if (live) {
        const F* val = (const F*)M.val;
        const u64 e = M.rowptr[i + 1];
        for (u64 k = M.rowptr[i] + lane; k < e; k += 64) {
            ZK_ASSERT_IDX(M.col[k] < m_vars);
            acc = fe_add(acc, fe_mul(val[k], z[M.col[k]]));
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 st = 32; st > 0; st >>= 1) {
        if (lane < st) {
            acc = fe_add(acc, sh[threadIdx.x + st]);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (live && lane == 0) (which == 0 ? oa : which == 1 ? ob : oc)[i] = acc;
}
// one WORKGROUP of 512 work-items per row of more than MATVEC_HUGE terms.  The stdlib SHA-256 circuit has rows of 7 041 terms; a wavefront
// walks one in 110 dependent rounds of (gather, product, sum) — 0.54 ms, and the witness map of a 4 ms proof waits for it
// (profiles/r7a_*: `k_matvec_long` 544 us) — a workgroup in 14.
static constexpr u32 MATVEC_HUGE = 512, MATVEC_HUGE_THREADS = 512;
template <class F>
__global__ void __launch_bounds__(MATVEC_HUGE_THREADS) k_matvec_huge(CsrDev A, CsrDev B, CsrDev C, const F* __restrict__ z, F* __restrict__ oa, F* __restrict__ ob,
                                                                     F* __restrict__ oc, const u64* __restrict__ rows, u64 m_vars) {
    ZK_PRIO_HIGH();
    __shared__ F sh[MATVEC_HUGE_THREADS];
    const u64 r = rows[blockIdx.x];
    const int which = (int)(r >> 32);
    const u64 i = r & 0xffffffffu;
    const CsrDev M = which == 0 ? A : which == 1 ? B : C;
    const F* val = (const F*)M.val;
    F acc = F::zero();
    const u64 e = M.rowptr[i + 1];
    for (u64 k = M.rowptr[i] + threadIdx.x; k < e; k += MATVEC_HUGE_THREADS) {
        ZK_ASSERT_IDX(M.col[k] < m_vars);
        acc = fe_add(acc, fe_mul(val[k], z[M.col[k]]));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 st = MATVEC_HUGE_THREADS >> 1; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            acc = fe_add(acc, sh[threadIdx.x + st]);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) (which == 0 ? oa : which == 1 ? ob : oc)[i] = acc;
}
// lanes per row for a matrix with `nnz` entries in `n` rows: the largest power of two <= half the average row length
static inline int matvec_group(u64 nnz, u64 n) {
    const u64 avg = n ? nnz / n : 0;
    int g = 1;
    while (g < 64 && (u64)g * 4 <= avg) g *= 2;
    return g;
}

// ---------------- LDS-resident sub-NTT ----------------
// An element is nine 29-bit limbs; limb l of slot s lives at lds[l * PL + s] (nine planes), so consecutive work-items
// touch consecutive 4-byte words of a plane.  Slots are padded by one per 32 elements (the radix-4 butterflies of the
// late rounds stride by 4, 16, 64 elements) and every sequence starts SS slots after the previous one.
static __device__ __forceinline__ int ntt_slot(int e) { return e + (e >> 5); }
static inline int ntt_seq_stride(int n) { return n + (n >> 5) + 1; }
template <class P>
__device__ __forceinline__ Fu<P> lds_get_u(const u32* lds, int PL, int slot) {
    Fu<P> r;
    ZK_UNROLL for (int l = 0; l < Fu<P>::N; ++l) r.v[l] = lds[l * PL + slot];
    return r;
}
template <class P>
__device__ __forceinline__ void lds_put_u(u32* lds, int PL, int slot, const Fu<P>& x) {
    ZK_UNROLL for (int l = 0; l < Fu<P>::N; ++l) lds[l * PL + slot] = x.v[l];
}

// Twiddle plan of an n-point sub-NTT: the factors in exactly the order the butterflies consume them, limb-major
// (limb l of entry i at plan[l * plen + i]), so that a wavefront's 64 work-items read 64 consecutive words per limb.
// Entries: for every radix-4 round with sub-length L > 4 (L = n, n/4, ...), q = L/4 values of w_L^pos, then q of w_L^(2 pos),
// then q of w_L^(3 pos); the last entry is w_4 = w_n^(n/4).
static inline u32 ntt_plan_len(int logn) {
    u32 len = 1;
    for (int L = 1 << logn; L > 4; L >>= 2) len += 3 * (u32)(L >> 2);
    return len;
}
// src[i] = exponent e of entry i (the entry is w_n^e); built on the host, gathered from the root table on the device
static inline void ntt_plan_exponents(int logn, std::vector<u32>& src) {
    const u32 n = 1u << logn;
    src.clear();
    for (u32 L = n; L > 4; L >>= 2) {
        const u32 q = L >> 2, tws = n / L;
        for (u32 k = 1; k <= 3; ++k)
            for (u32 pos = 0; pos < q; ++pos) src.push_back(k * pos * tws);
    }
    src.push_back(n >> 2);
}
// Entry-major: entry i = NTT_PLAN_STRIDE consecutive words (the nine limbs and padding to three 16-byte loads) — one address
// computation and three loads per factor.  (Round 5 kept the plan limb-major: nine 4-byte loads per factor, each with a 64-bit
// address of its own — 27 loads and as many address additions per butterfly; the plan of a 1024-point sub-NTT is 48 KiB either way
// and lives in the caches.)
static constexpr u32 NTT_PLAN_STRIDE = 12;
template <class P>
__global__ void k_ntt_plan_gather(const Fu<P>* __restrict__ roots, int rstride, const u32* __restrict__ src, u32 plen, u32* __restrict__ plan) {
    static_assert(Fu<P>::N <= (int)NTT_PLAN_STRIDE, "a plan entry holds the limbs of one factor");
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plen) return;
    const Fu<P> w = roots[(size_t)src[i] * rstride];
    ZK_UNROLL for (int l = 0; l < (int)NTT_PLAN_STRIDE; ++l) plan[(size_t)i * NTT_PLAN_STRIDE + l] = l < Fu<P>::N ? w.v[l] : 0u;
}
template <class P>
__device__ __forceinline__ Fu<P> ntt_plan_get(const u32* __restrict__ plan, u32 plen, u32 i) {
    (void)plen;
    const uint4* e = (const uint4*)(plan + (size_t)i * NTT_PLAN_STRIDE);
    const uint4 a = e[0], b = e[1], c = e[2];
    const u32 w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    Fu<P> r;
    ZK_UNROLL for (int l = 0; l < Fu<P>::N; ++l) r.v[l] = w[l];
    return r;
}

// The product of the passes: one operand a twiddle / element-wise factor straight from its table (TIGHT, limbs < 2^B), the other
// TIGHT or a lazily added / subtracted pair (limbs < 2^(B+2)) — the operand shapes the loose quotient digits of fieldu.cuh are
// sized for (fu_mul_loose: eight masks fewer per product; UConst::LOOSE_OK checked at compile time, the columns in the
// ZK_CHECK_OVERFLOW builds).
#ifndef ZK_NTT_LOOSE
#define ZK_NTT_LOOSE 1
#endif
template <class P>
__device__ __forceinline__ Fu<P> fu_mul_ntt(const Fu<P>& a, const Fu<P>& b) {
    return ZK_NTT_LOOSE ? fu_mul_loose(a, b) : fu_mul_inl(a, b);
}
// In-place DIF over `nseq` sequences of n = 2^logn points; the result is left in bit-reversed index order.
// Radix-4 butterflies = two radix-2 stages each:
//   (a, b, c, d) at i, i+q, i+2q, i+3q (q = L/4)  ->  a+b+c+d | (a-b+c-d) w^2pos | (a-c + w4(b-d)) w^pos | (a-c - w4(b-d)) w^3pos
// Value bounds (fieldu.cuh): inputs < 2p; sums/differences < 12p; products < 2p; the untwiddled output is brought back
// below 2p by fe_relax.  The last round of an even-length transform has trivial twiddles and leaves values < 12p, which
// the store path of the passes accepts.
// (first_done: the round of length n has been done on the way in — ntt_first_round below — and the data in LDS is its output)
template <class P>
__device__ __forceinline__ void lds_ntt_dif4(u32* lds, int PL, int SS, int logn, int nseq, const u32* __restrict__ plan, u32 plen, bool first_done = false) {
    typedef Fu<P> U;
    const int n = 1 << logn;
    const int T = blockDim.x;
    if (logn >= 2) {
        const int per_seq = n >> 2, nbf = nseq * per_seq;
        u32 off = first_done ? 3u * (u32)(n >> 2) : 0u;
        int logq = first_done ? logn - 4 : logn - 2;                      // log2 of q = L / 4 (every length is a power of two: shifts, no divisions)
        for (int L = first_done ? n >> 2 : n; L >= 4; L >>= 2, logq -= 2) {
            const int q = L >> 2;
            for (int b = threadIdx.x; b < nbf; b += T) {
                const int seq = b >> (logn - 2), bb = b & (per_seq - 1);
                const int grp = bb >> logq, pos = bb & (q - 1);
                const int base = seq * SS, i0 = grp * L + pos;
                const int s0 = base + ntt_slot(i0), s1 = base + ntt_slot(i0 + q), s2 = base + ntt_slot(i0 + 2 * q), s3 = base + ntt_slot(i0 + 3 * q);
                // ordered to keep few values alive: (b, d) first, each output stored as soon as it exists
                U t2, t3;
                {
                    const U bq = lds_get_u<P>(lds, PL, s1), d = lds_get_u<P>(lds, PL, s3);
                    t2 = fe_add(bq, d);
                    t3 = fu_mul_ntt<P>(fe_sub_k_lazy<4>(bq, d), ntt_plan_get<P>(plan, plen, plen - 1));   // * w4 (a difference that is
                                                                                  // multiplied at once skips its carry round: fieldu.cuh)
                }
                U t0, t1;
                {
                    const U a = lds_get_u<P>(lds, PL, s0), c = lds_get_u<P>(lds, PL, s2);
                    t0 = fe_add(a, c);
                    t1 = fe_sub_k<4>(a, c);
                }
                lds_put_u<P>(lds, PL, s0, fe_relax(fe_add(t0, t2)));
                if (L > 4) {
                    lds_put_u<P>(lds, PL, s1, fu_mul_ntt<P>(fe_sub_k_lazy<8>(t0, t2), ntt_plan_get<P>(plan, plen, off + q + pos)));
                    lds_put_u<P>(lds, PL, s2, fu_mul_ntt<P>(fe_add_lazy(t1, t3), ntt_plan_get<P>(plan, plen, off + pos)));
                    lds_put_u<P>(lds, PL, s3, fu_mul_ntt<P>(fe_sub_k_lazy<4>(t1, t3), ntt_plan_get<P>(plan, plen, off + 2 * q + pos)));
                } else {
                    lds_put_u<P>(lds, PL, s1, fe_sub_k<8>(t0, t2));
                    lds_put_u<P>(lds, PL, s2, fe_add(t1, t3));
                    lds_put_u<P>(lds, PL, s3, fe_sub_k<4>(t1, t3));
                }
            }
            off += 3 * (u32)q;
            __syncthreads();
        }
    }
    if (logn & 1) {   // the last radix-2 stage (pairs of neighbours, no twiddle)
        const int per_seq = n >> 1, nb2 = nseq * per_seq;
        for (int b = threadIdx.x; b < nb2; b += T) {
            const int seq = b >> (logn - 1), k = b & (per_seq - 1);
            const int s0 = seq * SS + ntt_slot(2 * k), s1 = seq * SS + ntt_slot(2 * k + 1);
            const U u = lds_get_u<P>(lds, PL, s0), v = lds_get_u<P>(lds, PL, s1);
            lds_put_u<P>(lds, PL, s0, fe_add(u, v));
            lds_put_u<P>(lds, PL, s1, fe_sub_k<4>(u, v));
        }
        __syncthreads();
    }
}
// The FIRST radix-4 round of a sub-transform (length n, q = n / 4), done on the four elements a work-item has just fetched —
// positions pos, pos + q, pos + 2q, pos + 3q of sequence `seq` are exactly what a pass's load phase hands one work-item when the
// workgroup has a quarter as many work-items as the tile has elements — and written to LDS as that round's output: the tile skips
// one trip through LDS (nine stores and nine loads per element) and one barrier of the pass's seven.  Needs n >= 16 (a twiddled
// round); same arithmetic and bounds as the round in lds_ntt_dif4.
template <class P>
__device__ __forceinline__ void ntt_first_round(u32* lds, int PL, int base, int logn, int pos, const Fu<P>& a, const Fu<P>& bq, const Fu<P>& c, const Fu<P>& d,
                                                const u32* __restrict__ plan, u32 plen) {
    typedef Fu<P> U;
    const int q = 1 << (logn - 2);
    const int s0 = base + ntt_slot(pos), s1 = base + ntt_slot(pos + q), s2 = base + ntt_slot(pos + 2 * q), s3 = base + ntt_slot(pos + 3 * q);
    const U t2 = fe_add(bq, d);
    const U t3 = fu_mul_ntt<P>(fe_sub_k_lazy<4>(bq, d), ntt_plan_get<P>(plan, plen, plen - 1));
    const U t0 = fe_add(a, c);
    const U t1 = fe_sub_k<4>(a, c);
    lds_put_u<P>(lds, PL, s0, fe_relax(fe_add(t0, t2)));
    lds_put_u<P>(lds, PL, s1, fu_mul_ntt<P>(fe_sub_k_lazy<8>(t0, t2), ntt_plan_get<P>(plan, plen, (u32)(q + pos))));
    lds_put_u<P>(lds, PL, s2, fu_mul_ntt<P>(fe_add_lazy(t1, t3), ntt_plan_get<P>(plan, plen, (u32)pos)));
    lds_put_u<P>(lds, PL, s3, fu_mul_ntt<P>(fe_sub_k_lazy<4>(t1, t3), ntt_plan_get<P>(plan, plen, (u32)(2 * q + pos))));
}
static __device__ __forceinline__ int bitrev_n(int x, int logn) { return logn ? (int)(__brev((unsigned)x) >> (32 - logn)) : 0; }

// one element between HBM (32 packed bytes, R'-form, any value < 2^256) and the working form
template <class P>
__device__ __forceinline__ Fu<P> ntt_load(const uint4* __restrict__ data, size_t g) {
    const uint4 lo = data[2 * g], hi = data[2 * g + 1];
    const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fe_relax(fu_unpack<P>(w));
}
// The passes fetch in BATCHES: every work-item issues the loads of all (up to NTT_BATCH) elements it moves before it touches the
// first — one memory latency per phase instead of one per element (round 5's loops loaded, waited and used element by element: four
// elements per work-item, eight exposed latencies per workgroup between the tile's load, its factors and its store — a quarter of a
// wavefront's cycles parked, profiles/r5a_stall_*.md).
static constexpr int NTT_BATCH = 4;
template <class P>
__device__ __forceinline__ Fu<P> ntt_unpack_words(const uint4& lo, const uint4& hi) {
    const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fe_relax(fu_unpack<P>(w));
}
// as ntt_store below, the factor already in registers
template <class P>
__device__ __forceinline__ void ntt_store_with(uint4* __restrict__ data, size_t g, Fu<P> x, bool have_post, const uint4& plo, const uint4& phi, int canon,
                                               const uint4* __restrict__ minus, size_t pg) {
    if (have_post) {
        const u32 w[8] = {plo.x, plo.y, plo.z, plo.w, phi.x, phi.y, phi.z, phi.w};
        x = fu_mul_ntt<P>(x, fu_unpack<P>(w));
    } else {
        x = fe_relax(x);
    }
    Fe<P> o;
    fu_pack(x, o.v);
    if (canon) fe_reduce_once(o);
    if (minus) {            // canonical exit only: o - minus[g] mod p, both canonical integers (the c term of the quotient)
        const uint4 lo = minus[2 * pg], hi = minus[2 * pg + 1];
        Fe<P> m;
        m.v[0] = lo.x; m.v[1] = lo.y; m.v[2] = lo.z; m.v[3] = lo.w; m.v[4] = hi.x; m.v[5] = hi.y; m.v[6] = hi.z; m.v[7] = hi.w;
        o = fe_sub(o, m);
    }
    data[2 * g] = make_uint4(o.v[0], o.v[1], o.v[2], o.v[3]);
    data[2 * g + 1] = make_uint4(o.v[4], o.v[5], o.v[6], o.v[7]);
}
// x: TIGHT, value < 32p.  With a post factor the product is < 2p; without one fe_relax leaves < 2p; `canon` adds the
// final conditional subtraction (the Montgomery exit of the last transform must leave canonical integers: MSM digits).
// `pg`: index of this position's factor in `post` (and of its subtrahend in `minus`): the position counted from the start of the
// vector, masked for per-block tables — `g` counts from wherever `data` points.
template <class P>
__device__ __forceinline__ void ntt_store(uint4* __restrict__ data, size_t g, Fu<P> x, const uint4* __restrict__ post, int canon,
                                          const uint4* __restrict__ minus, size_t pg) {
    if (post) {
        const uint4 lo = post[2 * pg], hi = post[2 * pg + 1];
        const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        x = fu_mul_inl(x, fu_unpack<P>(w));
    } else {
        x = fe_relax(x);
    }
    Fe<P> o;
    fu_pack(x, o.v);
    if (canon) fe_reduce_once(o);
    if (minus) {            // canonical exit only: o - minus[g] mod p, both canonical integers (the c term of the quotient)
        const uint4 lo = minus[2 * pg], hi = minus[2 * pg + 1];
        Fe<P> m;
        m.v[0] = lo.x; m.v[1] = lo.y; m.v[2] = lo.z; m.v[3] = lo.w; m.v[4] = hi.x; m.v[5] = hi.y; m.v[6] = hi.z; m.v[7] = hi.w;
        o = fe_sub(o, m);
    }
    data[2 * g] = make_uint4(o.v[0], o.v[1], o.v[2], o.v[3]);
    data[2 * g + 1] = make_uint4(o.v[4], o.v[5], o.v[6], o.v[7]);
}

// Which (tile, vector) a workgroup of a pass takes.  A launch over TWO vectors (a and b of a proof go through every pass together)
// reads the same element-wise factors for both: a workgroup's tile of the `post` table is as large as its tile of the data.  The
// hardware deals consecutive workgroups round-robin over the 8 XCDs, each with an L2 of its own — in (x = tile, y = vector) order
// the two users of a table tile are half a launch apart and the second read comes from HBM again (counted: 1.5 x the data).  So the
// linear workgroup number is re-read as 16-blocks of (8 tiles x 2 vectors): workgroups L and L + 8 — the same XCD, back to back —
// take the same tile of the two vectors, and the second finds the factors in that XCD's L2.  A speed-only affinity: any mapping
// computes the same result.  (nvec != 2, or a tile count that is not a multiple of 8: the identity.)
static __device__ __forceinline__ void ntt_tile_of(u32& tile, u32& vec) {
    tile = blockIdx.x;
    vec = blockIdx.y;
    if (gridDim.y == 2 && (gridDim.x & 7u) == 0) {
        const u32 L = blockIdx.x + gridDim.x * blockIdx.y;
        const u32 g = L >> 4, r = L & 15u;
        tile = g * 8 + (r & 7u);
        vec = r >> 3;
    }
}
// Start skew.  A pass is load -> five butterfly rounds -> store per workgroup, and a launch over a 2^20 domain is exactly TWO rounds
// of workgroups (the machine's LDS holds half the elements): dispatched together, every workgroup of a round loads at the same time
// (a 32 MiB burst: ~7 us with the multipliers idle), computes at the same time (HBM idle), stores at the same time — the memory
// phases and the arithmetic of a pass add up instead of overlapping (46 us of VALU issue + 18 us of HBM time = the 64 us per
// pass-vector of rounds 3-5).  So the workgroups of the FIRST round (linear number < first_round) wait a pseudo-random time below
// `skew_ticks` (wall clock, 10 ns) before they load: co-resident workgroups drift apart, one's loads land under the other's
// butterflies, and the drift carries into the second round.  A scheduling aid: 0 = off (ZKHIP_TUNE_NTT_SKEW_US), no effect on results.
struct NttSkew { u32 ticks, first_round; };
static __device__ __forceinline__ void ntt_start_skew(NttSkew sk) {
#ifndef ZK_EMU
    if (!sk.ticks) return;
    const u32 L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (L >= sk.first_round) return;
    const u32 d = ((L * 2654435761u) >> 12) % sk.ticks;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(4);
#else
    (void)sk;
#endif
}
// "cols" pass: the matrix is n1 x n2 row-major; this workgroup owns columns [c0, c0 + C).  grid.y = vectors
// (consecutive vectors are vec_stride elements apart).
// workgroups of 512 work-items per CU the transform kernels are compiled for (4: 64 VGPRs, a few spilled; 3: 85 VGPRs)
#ifndef ZK_NTT_WGS_PER_CU
#define ZK_NTT_WGS_PER_CU 4
#endif
// blockIdx.z = batch: `data` advances by batch_stride elements per batch (the middle pass of a three-pass transform runs the
// n1 x n2 sub-matrix of every outer index as one batch entry).  The element-wise factor of position g (counted from the start
// of the vector) is post[g & post_mask]: all ones for a table as long as the vector, block length - 1 for a table per block.
template <class P>
__global__ void __launch_bounds__(512, ZK_NTT_WGS_PER_CU) k_ntt_cols(Fe<P>* __restrict__ data_, u64 vec_stride, int log_n1, u32 n2, int C, const u32* __restrict__ plan,
                                                     u32 plen, const Fe<P>* __restrict__ post_, int canon, const Fe<P>* __restrict__ minus_ = nullptr,
                                                     u64 batch_stride = 0, u64 post_mask = ~(u64)0, NttSkew skew = NttSkew{0, 0}, int fuse_first = 1) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    static_assert(P::N == 8, "Fr is 8 x 32-bit words");
    ntt_start_skew(skew);
    u32* lds = (u32*)smem;
    u32 tile, vec;
    ntt_tile_of(tile, vec);
    const size_t boff = (size_t)blockIdx.z * batch_stride;
    uint4* data = (uint4*)(data_ + (size_t)vec * vec_stride + boff);
    const uint4* post = (const uint4*)post_;
    const int n1 = 1 << log_n1;
    const int SS = n1 + (n1 >> 5) + 1, PL = C * SS;
    const u32 c0 = tile * C;
    const int total = C * n1;
    const int logC = 31 - __clz(C);                 // (C is a power of two)
    // a quarter as many work-items as elements: what a work-item fetches — rows k, k + n1/4, k + 2 n1/4, k + 3 n1/4 of one column — is
    // one butterfly of the first round, which is then done on the way in (ntt_first_round)
    const bool fused = fuse_first && total == NTT_BATCH * (int)blockDim.x && log_n1 >= 4;
    for (int e0 = threadIdx.x; e0 < total; e0 += NTT_BATCH * blockDim.x) {
        uint4 lo[NTT_BATCH] = {}, hi[NTT_BATCH] = {};
        ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
            const int e = e0 + q * blockDim.x;
            if (e < total) {
                const size_t g = (size_t)(e >> logC) * n2 + c0 + (e & (C - 1));
                lo[q] = data[2 * g];
                hi[q] = data[2 * g + 1];
            }
        }
        if (fused) {
            ntt_first_round<P>(lds, PL, (e0 & (C - 1)) * SS, log_n1, e0 >> logC, ntt_unpack_words<P>(lo[0], hi[0]), ntt_unpack_words<P>(lo[1], hi[1]),
                               ntt_unpack_words<P>(lo[2], hi[2]), ntt_unpack_words<P>(lo[3], hi[3]), plan, plen);
        } else {
            ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < total) lds_put_u<P>(lds, PL, (e & (C - 1)) * SS + ntt_slot(e >> logC), ntt_unpack_words<P>(lo[q], hi[q]));
            }
        }
    }
    __syncthreads();
    lds_ntt_dif4<P>(lds, PL, SS, log_n1, C, plan, plen, fused);
    for (int e0 = threadIdx.x; e0 < total; e0 += NTT_BATCH * blockDim.x) {
        uint4 plo[NTT_BATCH] = {}, phi[NTT_BATCH] = {};
        if (post) {
            ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < total) {
                    const size_t pg = (boff + (size_t)(e >> logC) * n2 + c0 + (e & (C - 1))) & post_mask;
                    plo[q] = post[2 * pg];
                    phi[q] = post[2 * pg + 1];
                }
            }
        }
        ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
            const int e = e0 + q * blockDim.x;
            if (e < total) {
                const int k = e >> logC, j = e & (C - 1);
                const size_t g = (size_t)k * n2 + c0 + j;
                ntt_store_with<P>(data, g, lds_get_u<P>(lds, PL, j * SS + ntt_slot(bitrev_n(k, log_n1))), post != nullptr, plo[q], phi[q], canon, (const uint4*)minus_,
                                  (boff + g) & post_mask);
            }
        }
    }
}

// "rows" pass: this workgroup owns rows [r0, r0 + R) of n2 contiguous elements each.
template <class P>
__global__ void __launch_bounds__(512, ZK_NTT_WGS_PER_CU) k_ntt_rows(Fe<P>* __restrict__ data_, u64 vec_stride, int log_n2, int R, const u32* __restrict__ plan, u32 plen,
                                                     const Fe<P>* __restrict__ post_, int canon, const Fe<P>* __restrict__ minus_ = nullptr,
                                                     u64 post_mask = ~(u64)0, NttSkew skew = NttSkew{0, 0}, int fuse_first = 1) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    ntt_start_skew(skew);
    u32* lds = (u32*)smem;
    u32 tile, vec;
    ntt_tile_of(tile, vec);
    uint4* data = (uint4*)(data_ + (size_t)vec * vec_stride);
    const uint4* post = (const uint4*)post_;
    const int n2 = 1 << log_n2;
    const int SS = n2 + (n2 >> 5) + 1, PL = R * SS;
    const size_t base = (size_t)tile * R * n2;
    const int total = R * n2;
    // (as in the cols pass: with a quarter as many work-items as elements, work-item t fetches positions pos + k n2/4 of row t / (n2/4)
    // — one butterfly of the first round — and does it on the way in)
    const bool fused = fuse_first && total == NTT_BATCH * (int)blockDim.x && log_n2 >= 4;
    if (fused) {
        const int seq = (int)threadIdx.x >> (log_n2 - 2), pos = (int)threadIdx.x & ((n2 >> 2) - 1);
        uint4 lo[NTT_BATCH], hi[NTT_BATCH];
        ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
            const size_t g = base + (size_t)seq * n2 + pos + (size_t)q * (n2 >> 2);
            lo[q] = data[2 * g];
            hi[q] = data[2 * g + 1];
        }
        ntt_first_round<P>(lds, PL, seq * SS, log_n2, pos, ntt_unpack_words<P>(lo[0], hi[0]), ntt_unpack_words<P>(lo[1], hi[1]), ntt_unpack_words<P>(lo[2], hi[2]),
                           ntt_unpack_words<P>(lo[3], hi[3]), plan, plen);
    } else {
        for (int e0 = threadIdx.x; e0 < total; e0 += NTT_BATCH * blockDim.x) {
            uint4 lo[NTT_BATCH] = {}, hi[NTT_BATCH] = {};
            ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < total) {
                    lo[q] = data[2 * (base + e)];
                    hi[q] = data[2 * (base + e) + 1];
                }
            }
            ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < total) lds_put_u<P>(lds, PL, (e >> log_n2) * SS + ntt_slot(e & (n2 - 1)), ntt_unpack_words<P>(lo[q], hi[q]));
            }
        }
    }
    __syncthreads();
    lds_ntt_dif4<P>(lds, PL, SS, log_n2, R, plan, plen, fused);
    for (int e0 = threadIdx.x; e0 < total; e0 += NTT_BATCH * blockDim.x) {
        uint4 plo[NTT_BATCH] = {}, phi[NTT_BATCH] = {};
        if (post) {
            ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
                const int e = e0 + q * blockDim.x;
                if (e < total) {
                    const size_t pg = (base + e) & post_mask;
                    plo[q] = post[2 * pg];
                    phi[q] = post[2 * pg + 1];
                }
            }
        }
        ZK_UNROLL for (int q = 0; q < NTT_BATCH; ++q) {
            const int e = e0 + q * blockDim.x;
            if (e < total) {
                const int r = e >> log_n2, k = e & (n2 - 1);
                ntt_store_with<P>(data, base + e, lds_get_u<P>(lds, PL, r * SS + ntt_slot(bitrev_n(k, log_n2))), post != nullptr, plo[q], phi[q], canon,
                                  (const uint4*)minus_, (base + e) & post_mask);
            }
        }
    }
}

}  // namespace zk

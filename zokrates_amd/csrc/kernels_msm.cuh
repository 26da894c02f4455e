// kernels_msm.cuh — Pippenger multi-scalar multiplication on the device (K5-K8).
//
// Device replacement for [UPSTREAM] `VariableBaseMSM::multi_scalar_mul` (ark-ec 0.3.0, SURVEY.md
// App. A.5), the five calls of which dominate `Groth16::prove`
// (/root/reference/zokrates_ark/src/groth16.rs:44).  Same mathematics (bucket method), different
// schedule: instead of one CPU thread per window walking all scalars, the device
//   1. recodes every scalar into W signed c-bit digits (half the buckets: K = 2^(c-1) per set; zero digits are dropped, so
//      a scalar 0 costs nothing and a scalar 1 is one entry of bucket 0),
//   2. counting-sorts the (bucket set, bucket) keys so that every bucket's points are contiguous,
//   3. accumulates with XYZZ mixed additions in a *load-balanced* way: the sorted list is cut into as many equal slices
//      as the machine holds work-items, whatever buckets they cover, and every work-item emits one partial sum per
//      bucket it touches (a segmented reduction; slot = key + lane is collision free),
//   4. buckets that ended up spread over many slices are reduced by a whole workgroup each,
//   5. folds each set's buckets (sum of (b+1) B_b) in a two-digit form with scan tails.
// Resident keys carry PRECOMPUTED WINDOW MULTIPLES: table level j holds 2^(c j) P for every base P, so digit j of a scalar
// pairs with level j of its base and ALL windows share ONE bucket set ("shared" mode: one fold per MSM instead of W, no
// Horner step, buckets hundreds of points deep).  Ad-hoc bases (zkhip_msm_g1/g2) have no table: W bucket sets and a
// host Horner step over the W window sums, through the same kernels.
// One digit/sort pass is shared by every base set that uses the same scalars (a_query, b_g1_query,
// b_g2_query and l_query all pair with z).  The result is the exact group element, so it is
// independent of c, of the slice length, of the summation order and of how ark itself schedules the sum.
#pragma once
#include "devrt.h"
#include "ec.cuh"

namespace zk {

// per point type tuning.  ACCUM_WPE: waves per SIMD the accumulation kernel's registers must leave room for; SLICE_WPE / FUSED_WPE:
// how finely the sorted list is cut — slices per SIMD lane of a launch over one table / over the tables that share a list.
// Round 5 (profiles/r5a_*, r5b_*: same-box sessions): with the general path of the addition out of line (ec.cuh
// xyzz_madd_cold) the hot loop needs 128 registers for G1 and 256 for G2 without a spill — FOUR waves per SIMD for G1, TWO for G2
// (round 4: 168 registers = three waves with four scratch accesses per step; 427 = one wave, which cannot issue more than three
// cycles in four).  More slices than resident waves: the launch then runs in rounds and the dispatcher evens out what one round of
// equal slices cannot (CUs do not all run at the same speed): G1 over three tables 5.43 ms at 5 slices per lane, 4.96 at 6, 5.15 at
// 8, 5.78 at 4 (= one round); G2 3.38 ms at 2, 3.30 at 4.
#ifndef ZK_G2_ACCUM_WPE
#define ZK_G2_ACCUM_WPE 2
#endif
#ifndef ZK_G1_FUSED_WPE
#define ZK_G1_FUSED_WPE 6
#endif
#ifndef ZK_G1_SLICE_WPE
#define ZK_G1_SLICE_WPE 4
#endif
#ifndef ZK_G2_SLICE_WPE
#define ZK_G2_SLICE_WPE 4
#endif
// COLD_WPE: the register budget of the fold / heavy-bucket kernels, whose waves sit BESIDE the accumulation's: a fold wave of 361
// registers (round 4's G2 fold) leaves a SIMD room for one G1 accumulation wave and no G2 one for as long as it runs.  With the
// doubling of their general addition out of line (ec.cuh ZK_FOLD_DBL_CALL) they get by with the accumulation's own budgets:
// same box, 101-103 proofs/s at 3 / 1 waves per SIMD, 105-106 at 4 / 2 (profiles/r5f_fold_register_budgets_ab.txt).
#ifndef ZK_G2_COLD_WPE
#define ZK_G2_COLD_WPE 2
#endif
#ifndef ZK_G2_PREFETCH_REGS
#define ZK_G2_PREFETCH_REGS 1
#endif
#ifndef ZK_G2_ZZ_LDS
#define ZK_G2_ZZ_LDS 1
#endif
// PREFETCH_REGS: the next base travels in registers (else it is only touched one entry ahead and loaded where it is used)
#ifndef ZK_G1_ZZ_LDS
#define ZK_G1_ZZ_LDS 0
#endif
#ifndef ZK_G2_XY_LDS
#define ZK_G2_XY_LDS 0
#endif
#ifndef ZK_G1_ACCUM_WPE
#define ZK_G1_ACCUM_WPE 4
#endif
#ifndef ZK_G1_COLD_WPE
#define ZK_G1_COLD_WPE 4
#endif
template <class F> struct MsmTuning { static constexpr int ACCUM_WPE = ZK_G1_ACCUM_WPE, COLD_WPE = ZK_G1_COLD_WPE, FUSED_WPE = ZK_G1_FUSED_WPE, SLICE_WPE = ZK_G1_SLICE_WPE; static constexpr bool IS_EXT = false, PREFETCH_REGS = true, ZZ_IN_LDS = ZK_G1_ZZ_LDS != 0, XY_IN_LDS = false; };
template <class P_> struct MsmTuning<Fe2<P_>> { static constexpr int ACCUM_WPE = 2, COLD_WPE = 2, FUSED_WPE = 2, SLICE_WPE = 2; static constexpr bool IS_EXT = true, PREFETCH_REGS = true, ZZ_IN_LDS = false, XY_IN_LDS = false; };
template <class P_> struct MsmTuning<Fu2<P_>> { static constexpr int ACCUM_WPE = ZK_G2_ACCUM_WPE, COLD_WPE = ZK_G2_COLD_WPE, FUSED_WPE = ZK_G2_SLICE_WPE, SLICE_WPE = ZK_G2_SLICE_WPE; static constexpr bool IS_EXT = true, PREFETCH_REGS = ZK_G2_PREFETCH_REGS != 0, ZZ_IN_LDS = ZK_G2_ZZ_LDS != 0, XY_IN_LDS = ZK_G2_XY_LDS != 0; };
#ifndef ZK_BLS_G1_ACCUM_WPE
#define ZK_BLS_G1_ACCUM_WPE 2
#endif
#ifndef ZK_BLS_G2_ACCUM_WPE
#define ZK_BLS_G2_ACCUM_WPE 1
#endif
#ifndef ZK_BLS_G1_FUSED_WPE
#define ZK_BLS_G1_FUSED_WPE 3
#endif
#ifndef ZK_BLS_G1_SLICE_WPE
#define ZK_BLS_G1_SLICE_WPE 2
#endif
#ifndef ZK_BLS_G2_SLICE_WPE
#define ZK_BLS_G2_SLICE_WPE 1
#endif
#ifndef ZK_BLS_G2_ZZ_LDS
#define ZK_BLS_G2_ZZ_LDS 0
#endif
template <> struct MsmTuning<Fu<Bls381Fq>> { static constexpr int ACCUM_WPE = ZK_BLS_G1_ACCUM_WPE, COLD_WPE = 3, FUSED_WPE = ZK_BLS_G1_FUSED_WPE, SLICE_WPE = ZK_BLS_G1_SLICE_WPE; static constexpr bool IS_EXT = false, PREFETCH_REGS = true, ZZ_IN_LDS = false, XY_IN_LDS = false; };
template <> struct MsmTuning<Fu2<Bls381Fq>> { static constexpr int ACCUM_WPE = ZK_BLS_G2_ACCUM_WPE, COLD_WPE = 1, FUSED_WPE = ZK_BLS_G2_SLICE_WPE, SLICE_WPE = ZK_BLS_G2_SLICE_WPE; static constexpr bool IS_EXT = true, PREFETCH_REGS = true, ZZ_IN_LDS = ZK_BLS_G2_ZZ_LDS != 0, XY_IN_LDS = false; };
// dynamic LDS of one accumulation workgroup (256 work-items): the coordinates of the running sum that live there, word-major
template <class F> constexpr size_t msm_accum_lds_bytes() { return (size_t)((MsmTuning<F>::ZZ_IN_LDS ? 2 : 0) + (MsmTuning<F>::XY_IN_LDS ? 2 : 0)) * (sizeof(F) / 4) * 256 * 4; }
// the base tables of the MSMs one launch serves (A, B1 and L of a proof share the sort of the assignment)
static constexpr int MSM_MAX_TABLES = 3;
struct MsmTables {
    const void* p[MSM_MAX_TABLES];
};
static constexpr u32 BIND_SHORT_COL = 16;   // bind.cuh: columns of W of at most this many entries are summed by one work-item (the host lists the others)
static constexpr u32 MSM_MIN_SLICE = 8;   // default for the finest cut of the sorted list (small inputs leave work-items idle)

// The kernels below work on points over the UNSATURATED field types of fieldu.cuh (Fu / Fu2); only the window sums
// leaving k_msm_fold_final go back to the saturated Montgomery form the host code uses.
template <class P_> ZK_HD Fe<P_> to_sat(const Fu<P_>& a) { return fu_to_fe(a); }
template <class P_> ZK_HD Fe2<P_> to_sat(const Fu2<P_>& a) { return fu_to_fe(a); }
template <class P_> ZK_HD Fe<P_> to_sat(const Fe<P_>& a) { return a; }
template <class P_> ZK_HD Fe2<P_> to_sat(const Fe2<P_>& a) { return a; }
template <class FS, class U>
ZK_HD Xyzz<FS> xyzz_to_sat(const Xyzz<U>& p) {
    if (p.is_inf()) return Xyzz<FS>::inf();
    return {to_sat(p.x), to_sat(p.y), to_sat(p.zz), to_sat(p.zzz)};
}
// ---- packed affine points: what the resident base tables hold ----
// x | y as packed integers (fu_pack: the value of x*R' mod p in 32-bit words), infinity all-zero.  A BN254 G1 point is one
// aligned 64-byte line (the 9 x 29-bit working form is 72 bytes and straddles two), G2 128 bytes; BLS12-381 96 / 192.
template <class F>
struct alignas(16) AffPacked {
    static constexpr int NW = PackedWords<F>::N;
    u32 w[2 * NW];
};
template <class F>
ZK_HD Aff<F> aff_unpack(const u32* w) {
    return {FuUnpack<F>::get(w), FuUnpack<F>::get(w + PackedWords<F>::N)};
}
template <class F>
ZK_HD void aff_pack(const Aff<F>& p, AffPacked<F>* out) {   // coordinates TIGHT with value < 2^(32 W) (products: < 2p)
    fu_pack(p.x, out->w);
    fu_pack(p.y, out->w + PackedWords<F>::N);
}
template <class F>
__device__ __forceinline__ void aff_load_words(const AffPacked<F>* __restrict__ tbl, u64 idx, u32* w) {
    const uint4* src = (const uint4*)(tbl + idx);
    ZK_UNROLL for (int q = 0; q < AffPacked<F>::NW / 2; ++q) {
        const uint4 t = src[q];
        w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
}
// affine points: saturated Montgomery -> packed working form (key load, level 0 of a table)
template <class FS, class U>
__global__ void k_points_to_packed(const Aff<FS>* __restrict__ in, AffPacked<U>* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<FS> p = in[i];
    Aff<U> q = Aff<U>::inf();
    if (!p.is_inf()) { q.x = fu_from_fe(p.x); q.y = fu_from_fe(p.y); }
    aff_pack(q, out + i);
}

static constexpr u32 MSM_NO_DIGIT = 0xffffffffu;
// work-items of a sort workgroup (one workgroup per CU: its histogram takes up to 128 KiB of LDS); 1024 measured the same as
// 512 (profiles/r3e_sort_workgroup_ab.txt)
#ifndef ZK_SORT_THREADS
#define ZK_SORT_THREADS 512
#endif
static constexpr u32 MSM_HEAVY = 16;  // a bucket spread over more slices than this is reduced by a whole workgroup (a value repeated
                                      // across a witness — the constant in every first S-box of a Poseidon chain, the ones of a
                                      // boolean-heavy assignment — would otherwise be summed serially by one work-item of the fold)

// ---- 1. signed-digit recoding, counting sort — window-major, histograms in LDS ----
// One scalar produces W digits; doing the sort with one global atomic per digit makes it atomic-bound (17.8 M atomics per
// pass at 2^20).  Here a workgroup owns ONE window of a chunk of scalars, so all its keys fall into one set of K buckets:
// the histogram lives in LDS (up to 128 KiB) and only one global atomic per touched (workgroup, bucket) remains.
// The digits are computed ONCE (k_msm_digits: one work-item per scalar walks its windows with the carry in a register) and laid
// down window-major, dig[j * n + i]: the count and place passes then read one coalesced word per entry — round 4 recomputed every
// digit from the scalar's words in each of the three passes over it (and once more per half with 17-bit windows), four dependent
// global loads per entry at two waves per SIMD, which is what those kernels' time was (0.6 ms per sort at c = 17; now ~0.2).
// Window j belongs to bucket set j % sets and pairs with table level j / sets (level t holds 2^(c sets t) P): sets = 1 — every
// window multiple precomputed, ONE bucket set; sets = W — no table, a bucket set per window; in between, keys too large for
// W levels of every base (domains above 2^24) keep every sets-th multiple and fold `sets` bucket sets.
// key = (j % sets) * K + bucket;  sorted entry = ((j / sets) * idx_stride + i) | sign << 31 (idx_stride = table level stride).
// Signed digit of window j: bucket (|d| - 1) | sign << 31, or MSM_NO_DIGIT.  digit = raw + carry_in, minus 2^c (and a carry out)
// when that exceeds K = 2^(c-1); zero digits are dropped (a scalar 0 costs nothing, a scalar 1 is one entry of bucket 0).
// (It also zeroes the counters the passes behind it accumulate into — `za` / `zb` / `zc`, any of them may be null — instead of three
// fill launches of their own: every launch of a sort that runs beside an accumulation waits for a place of its own, 35-300 us each
// in a lone proof's trace, profiles/r6h_*gantt*.)
struct MsmZero { u32* p[3]; u32 n[3]; };
static __device__ __forceinline__ void msm_zero(const MsmZero& z) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride = (u64)gridDim.x * blockDim.x;
    ZK_UNROLL for (int k = 0; k < 3; ++k)
        if (z.p[k]) for (u64 i = g; i < z.n[k]; i += stride) z.p[k][i] = 0;
}
static __global__ void k_msm_zero(MsmZero z) { ZK_PRIO_HIGH(); msm_zero(z); }
static __global__ void k_msm_digits(const u32* __restrict__ scalars, u64 n, int c, int W, u32* __restrict__ dig, MsmZero zero) {
    ZK_PRIO_HIGH();
    msm_zero(zero);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* __restrict__ sw = scalars + i * 8;        // 8 canonical words (their 32 bytes stay in the cache across the windows)
    const u32 K = 1u << (c - 1), full = 1u << c;
    u32 carry = 0;
    for (int j = 0; j < W; ++j) {
        const int bit = j * c;
        u32 raw = 0;
        if (bit < 256) {
            const int limb = bit >> 5, offb = bit & 31;
            u64 two = sw[limb];
            if (offb + c > 32 && limb + 1 < 8) two |= (u64)sw[limb + 1] << 32;
            raw = (u32)(two >> offb) & (full - 1);
        }
        raw += carry;
        u32 d;
        if (raw > K) {
            const u32 mag = full - raw;                   // raw == 2^c (an all-ones window plus the carry) is digit 0 with a carry out
            d = mag ? ((mag - 1) | 0x80000000u) : MSM_NO_DIGIT;
            carry = 1;
        } else {
            d = raw ? raw - 1 : MSM_NO_DIGIT;
            carry = 0;
        }
        dig[(u64)j * n + i] = d;
    }
}
// `keep` (may be null: everything takes part): bit i CLEAR = scalar i takes no part — the sort of the B-family MSMs leaves out the
// variables whose bases are the point at infinity (zkhip_pk::thin_keep).
static __device__ __forceinline__ bool msm_skipped(const u32* __restrict__ keep, u64 i) { return keep && !((keep[i >> 5] >> (i & 31)) & 1u); }
static constexpr int MSM_SORT_ILP = 4;     // entries a work-item of the sort has in flight per round (their loads are issued together)
// grid (nchunks, W, K / kh); dynamic LDS kh x 4 B (kh <= 2^15: a window of more than 16 bits comes in halves, blockIdx.z, each
// workgroup keeping the digits of its own range of buckets).  cnt[key] += number of digits with that key in this chunk.
static __global__ void __launch_bounds__(ZK_SORT_THREADS) k_msm_count(const u32* __restrict__ dig, u64 n, int c, int W, u64 chunk, u32 sets, u32 kh,
                                                        u32* __restrict__ cnt, const u32* __restrict__ keep) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    u32* hist = (u32*)smem;
    const u32 K = 1u << (c - 1);
    const int j = blockIdx.y;
    const u32 b0 = blockIdx.z * kh;
    const u32* __restrict__ dj = dig + (u64)j * n;
    for (u32 b = threadIdx.x; b < kh; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    for (u64 i = i0 + threadIdx.x; i < i1; i += (u64)MSM_SORT_ILP * blockDim.x) {
        u32 d[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u64 ii = i + (u64)q * blockDim.x;
            d[q] = (ii < i1 && !msm_skipped(keep, ii)) ? dj[ii] : MSM_NO_DIGIT;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            if (d[q] == MSM_NO_DIGIT) continue;
            const u32 b = (d[q] & 0x7fffffffu) - b0;
            if (b < kh) atomicAdd(&hist[b], 1u);
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < kh; b += blockDim.x)
        if (hist[b]) atomicAdd(&cnt[(u64)((u32)j % sets) * K + b0 + b], hist[b]);
}
// same geometry; after the scan: reserve this workgroup's run inside every bucket it touches (one global atomic per
// bucket, several in flight per work-item), then place the entries with LDS atomics.
static __global__ void __launch_bounds__(ZK_SORT_THREADS) k_msm_place(const u32* __restrict__ dig, u64 n, int c, int W, u64 chunk, u32 sets, u32 kh,
                                                        u64 idx_stride, const u32* __restrict__ off, u32* __restrict__ cursor,
                                                        u32* __restrict__ sorted, const u32* __restrict__ keep) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    u32* hist = (u32*)smem;
    const u32 K = 1u << (c - 1);
    const int j = blockIdx.y;
    const u32 b0 = blockIdx.z * kh;
    const u32* __restrict__ dj = dig + (u64)j * n;
    for (u32 b = threadIdx.x; b < kh; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    for (u64 i = i0 + threadIdx.x; i < i1; i += (u64)MSM_SORT_ILP * blockDim.x) {
        u32 d[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u64 ii = i + (u64)q * blockDim.x;
            d[q] = (ii < i1 && !msm_skipped(keep, ii)) ? dj[ii] : MSM_NO_DIGIT;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            if (d[q] == MSM_NO_DIGIT) continue;
            const u32 b = (d[q] & 0x7fffffffu) - b0;
            if (b < kh) atomicAdd(&hist[b], 1u);
        }
    }
    __syncthreads();
    const u64 key0 = (u64)((u32)j % sets) * K + b0;
    for (u32 b = threadIdx.x; b < kh; b += MSM_SORT_ILP * blockDim.x) {
        u32 have[MSM_SORT_ILP], start[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u32 bq = b + q * blockDim.x;
            have[q] = bq < kh ? hist[bq] : 0;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u32 bq = b + q * blockDim.x;
            start[q] = have[q] ? off[key0 + bq] + atomicAdd(&cursor[key0 + bq], have[q]) : 0;   // global position of this workgroup's first entry
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q)
            if (have[q]) hist[b + q * blockDim.x] = start[q];
    }
    __syncthreads();
    const u32 level = (u32)((u64)((u32)j / sets) * idx_stride);
    for (u64 i = i0 + threadIdx.x; i < i1; i += (u64)MSM_SORT_ILP * blockDim.x) {
        u32 d[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u64 ii = i + (u64)q * blockDim.x;
            d[q] = (ii < i1 && !msm_skipped(keep, ii)) ? dj[ii] : MSM_NO_DIGIT;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            if (d[q] == MSM_NO_DIGIT) continue;
            const u32 b = (d[q] & 0x7fffffffu) - b0;
            if (b >= kh) continue;
            const u32 pos = atomicAdd(&hist[b], 1u);
            ZK_ASSERT_IDX(pos >= off[key0 + b] && pos < off[key0 + b + 1]);
            sorted[pos] = (level + (u32)(i + (u64)q * blockDim.x)) | (d[q] & 0x80000000u);
        }
    }
}

// ---- 1b. the same placement in TWO levels (round 6; k_msm_place above stays as the path for windows below 9 bits) ----
// k_msm_place scatters 4-byte entries into runs of ~2 entries per (workgroup, bucket): 15.7 M stores, each to a line of its own —
// 548 MB of write traffic for 63 MB of payload, 0.28-0.36 ms per sort, behind a 128 KiB histogram that lets no second workgroup
// (and no accumulation workgroup with LDS of its own) share the CU.  Two levels:
//   k_msm_part_coarse   a workgroup = one window of a chunk of scalars, as before, but it only separates the 256-key COARSE bins
//                       (key >> 8; 1 KiB of counters): runs of ~60 entries per (workgroup, bin), written as 8-byte pairs
//                       (key & 255, entry) into the bin's region of `pairs` — the region every bucket of the bin will occupy in
//                       `sorted`, so the bucket offsets of the counting pass serve both levels;
//   k_msm_tile_offsets  how many tiles of MSM_FINE_TILE pairs every coarse region is cut into (one small workgroup);
//   k_msm_part_fine     a workgroup = one tile of one region: 256 counters, one global atomic per touched bucket, entries placed
//                       into runs of ~8 per (tile, bucket) inside a 0.25 MB region of `sorted` that the L2 keeps whole.
// Same result as k_msm_place up to the order of the entries inside a bucket, which no consumer depends on.
static constexpr u32 MSM_COARSE_BITS = 8;                 // keys per coarse bin = 256 (a window narrower than 9 bits takes k_msm_place)
static constexpr u32 MSM_FINE_TILE = 2048;                // pairs per workgroup of the fine pass (8 per work-item)
static __global__ void __launch_bounds__(256) k_msm_part_coarse(const u32* __restrict__ dig, u64 n, int c, int W, u64 chunk, u32 sets, u64 idx_stride,
                                                              const u32* __restrict__ off, u32* __restrict__ ccur, unsigned long long* __restrict__ pairs,
                                                              const u32* __restrict__ keep) {
    ZK_PRIO_HIGH();
    __shared__ u32 hist[256];
    const u32 K = 1u << (c - 1), nb = K >> MSM_COARSE_BITS;          // coarse bins of one bucket set (<= 256: K <= 2^16)
    const int j = blockIdx.y;
    const u32 set = (u32)j % sets;
    const u32* __restrict__ dj = dig + (u64)j * n;
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const u64 i0 = (u64)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    for (u64 i = i0 + threadIdx.x; i < i1; i += (u64)MSM_SORT_ILP * blockDim.x) {
        u32 d[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u64 ii = i + (u64)q * blockDim.x;
            d[q] = (ii < i1 && !msm_skipped(keep, ii)) ? dj[ii] : MSM_NO_DIGIT;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q)
            if (d[q] != MSM_NO_DIGIT) atomicAdd(&hist[(d[q] & 0x7fffffffu) >> MSM_COARSE_BITS], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < nb; b += blockDim.x) {
        const u32 have = hist[b], bin = set * nb + b;
        hist[b] = have ? off[(u64)bin << MSM_COARSE_BITS] + atomicAdd(&ccur[bin], have) : 0;   // this workgroup's run inside the bin's region
    }
    __syncthreads();
    const u32 level = (u32)((u64)((u32)j / sets) * idx_stride);
    for (u64 i = i0 + threadIdx.x; i < i1; i += (u64)MSM_SORT_ILP * blockDim.x) {
        u32 d[MSM_SORT_ILP];
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            const u64 ii = i + (u64)q * blockDim.x;
            d[q] = (ii < i1 && !msm_skipped(keep, ii)) ? dj[ii] : MSM_NO_DIGIT;
        }
        ZK_UNROLL for (int q = 0; q < MSM_SORT_ILP; ++q) {
            if (d[q] == MSM_NO_DIGIT) continue;
            const u32 b = d[q] & 0x7fffffffu;
            const u32 pos = atomicAdd(&hist[b >> MSM_COARSE_BITS], 1u);
            const u32 entry = (level + (u32)(i + (u64)q * blockDim.x)) | (d[q] & 0x80000000u);
            pairs[pos] = (unsigned long long)(b & 255u) << 32 | entry;
        }
    }
}
// tile_off[r] = tiles of the coarse regions before r (tile_off[nbins] = all of them); one workgroup, nbins <= a few thousand
static constexpr int MSM_TILE_SCAN_THREADS = 256;
static __global__ void k_msm_tile_offsets(const u32* __restrict__ off, u32 nbins, u32* __restrict__ tile_off) {
    ZK_PRIO_HIGH();
    __shared__ u32 sh[MSM_TILE_SCAN_THREADS];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < nbins; b0 += MSM_TILE_SCAN_THREADS) {
        const u32 r = b0 + threadIdx.x;
        u32 x = 0;
        if (r < nbins) {
            const u32 size = off[(u64)(r + 1) << MSM_COARSE_BITS] - off[(u64)r << MSM_COARSE_BITS];
            x = (size + MSM_FINE_TILE - 1) / MSM_FINE_TILE;
        }
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int d = 1; d < MSM_TILE_SCAN_THREADS; d <<= 1) {
            const u32 t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (r < nbins) tile_off[r] = carry + sh[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == MSM_TILE_SCAN_THREADS - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_off[nbins] = carry;
}
static __global__ void __launch_bounds__(256) k_msm_part_fine(const unsigned long long* __restrict__ pairs, const u32* __restrict__ off, const u32* __restrict__ tile_off,
                                                            u32 nbins, u32* __restrict__ cursor, u32* __restrict__ sorted) {
    ZK_PRIO_HIGH();
    __shared__ u32 hist[256];
    const u32 t = blockIdx.x;
    if (t >= tile_off[nbins]) return;                    // (uniform: the grid is an upper bound)
    u32 lo = 0, hi = nbins;                              // the region of tile t: tile_off[lo] <= t < tile_off[hi]
    while (hi - lo > 1) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (tile_off[mid] <= t) lo = mid; else hi = mid;
    }
    const u32 r = lo;
    const u32 rbeg = off[(u64)r << MSM_COARSE_BITS], rend = off[(u64)(r + 1) << MSM_COARSE_BITS];
    const u32 p0 = rbeg + (t - tile_off[r]) * MSM_FINE_TILE;
    const u32 p1 = rend - p0 > MSM_FINE_TILE ? p0 + MSM_FINE_TILE : rend;
    hist[threadIdx.x] = 0;
    __syncthreads();
    constexpr int PER = MSM_FINE_TILE / 256;
    unsigned long long e[PER];
    ZK_UNROLL for (int q = 0; q < PER; ++q) {
        const u32 p = p0 + threadIdx.x + (u32)q * 256u;
        e[q] = p < p1 ? pairs[p] : ~0ull;
    }
    ZK_UNROLL for (int q = 0; q < PER; ++q)
        if (e[q] != ~0ull) atomicAdd(&hist[(u32)(e[q] >> 32) & 255u], 1u);
    __syncthreads();
    {
        const u32 key = (r << MSM_COARSE_BITS) + threadIdx.x, have = hist[threadIdx.x];
        hist[threadIdx.x] = have ? off[key] + atomicAdd(&cursor[key], have) : 0;
    }
    __syncthreads();
    ZK_UNROLL for (int q = 0; q < PER; ++q) {
        if (e[q] == ~0ull) continue;
        const u32 b = (u32)(e[q] >> 32) & 255u;
        const u32 pos = atomicAdd(&hist[b], 1u);
        ZK_ASSERT_IDX(pos >= off[(r << MSM_COARSE_BITS) + b] && pos < off[(r << MSM_COARSE_BITS) + b + 1]);
        sorted[pos] = (u32)e[q];
    }
}

// ---- 2. exclusive scan of the counters in ONE workgroup (nk <= SCAN_ONE_MAX: every resident key has 2^16 of them) ----
// off[k] = counters before k, off[nk] = their sum; with `tile_off` (the two-level placement) also the tiles of every coarse bin —
// one launch where round 5 ran three (k_scan_local / _chunks / _add below: kept for more counters than one workgroup takes) and
// the two-level placement a fourth.  Tiles of 4096 counters: a work-item loads four consecutive ones (coalesced 16-byte loads), the
// wavefront scans its 64 sums with lane shuffles, the 16 wavefronts' totals meet in LDS, a running carry links the tiles.
static constexpr u32 SCAN_ONE_THREADS = 256, SCAN_ONE_PER = 16, SCAN_ONE_MAX = 1u << 17;
// (256 work-items: one wavefront per SIMD and 20 registers — a workgroup that finds a place beside accumulation kernels; a first
// version with 1024 work-items needed a whole CU to itself and waited 2.5 ms for one in a lone proof's trace)
static __device__ __forceinline__ u32 wave_inclusive_scan(u32 v) {
    const u32 lane = threadIdx.x & 63u;
    ZK_UNROLL for (u32 d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl(v, (int)(lane >= d ? lane - d : lane));
        if (lane >= d) v += o;
    }
    return v;
}
static __global__ void __launch_bounds__(256) k_scan_one(const u32* __restrict__ cnt, u32* __restrict__ off, u32 nk, u32* __restrict__ tile_off, u32 nbins) {
    ZK_PRIO_HIGH();
    constexpr u32 NW = SCAN_ONE_THREADS / 64, TILE = SCAN_ONE_THREADS * SCAN_ONE_PER;
    __shared__ u32 wsum[NW];
    __shared__ u32 coarse[(SCAN_ONE_MAX >> MSM_COARSE_BITS) + 1];      // off at every 256th key
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    u32 carry = 0;
    // (the tiles are a chain — carry, two barriers — but their LOADS are not: a work-item fetches the next tile's counters before it
    // scans this one's, so that a tile costs its scan and not a trip to memory as well; 16 tiles at 2^16 counters: 44 -> ~15 us alone,
    // and this kernel is on the path of every sort — the head of a lone proof, where nothing else can start before it)
    auto fetch = [&](u32 base, u32 (&v)[SCAN_ONE_PER]) {
        const u32 k = base + SCAN_ONE_PER * t;                         // this work-item's 16 consecutive counters
        if (k + SCAN_ONE_PER <= nk) {
            ZK_UNROLL for (u32 q = 0; q < SCAN_ONE_PER / 4; ++q) {
                const uint4 x = *(const uint4*)(cnt + k + 4 * q);
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
        } else {
            ZK_UNROLL for (u32 i = 0; i < SCAN_ONE_PER; ++i) v[i] = k + i < nk ? cnt[k + i] : 0;
        }
    };
    u32 nx[SCAN_ONE_PER];
    fetch(0, nx);
    for (u32 base = 0; base < nk; base += TILE) {
        const u32 k = base + SCAN_ONE_PER * t;
        u32 v[SCAN_ONE_PER];
        ZK_UNROLL for (u32 i = 0; i < SCAN_ONE_PER; ++i) v[i] = nx[i];
        if (base + TILE < nk) fetch(base + TILE, nx);
        u32 mine = 0;
        ZK_UNROLL for (u32 i = 0; i < SCAN_ONE_PER; ++i) mine += v[i];
        const u32 incl = wave_inclusive_scan(mine);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 before = carry, all = 0;
        ZK_UNROLL for (u32 w = 0; w < NW; ++w) {
            const u32 x = wsum[w];
            if (w < wave) before += x;
            all += x;
        }
        u32 run = before + incl - mine;
        static_assert((1u << MSM_COARSE_BITS) % SCAN_ONE_PER == 0, "a coarse bin starts at a work-item's first counter");
        if (tile_off && k < nk && (k & ((1u << MSM_COARSE_BITS) - 1)) == 0) coarse[k >> MSM_COARSE_BITS] = run;
        if (k + SCAN_ONE_PER <= nk) {                                  // (whole: four 16-byte stores)
            ZK_UNROLL for (u32 q = 0; q < SCAN_ONE_PER / 4; ++q) {
                uint4 o;
                o.x = run; run += v[4 * q];
                o.y = run; run += v[4 * q + 1];
                o.z = run; run += v[4 * q + 2];
                o.w = run; run += v[4 * q + 3];
                *(uint4*)(off + k + 4 * q) = o;
            }
        } else {
            ZK_UNROLL for (u32 i = 0; i < SCAN_ONE_PER; ++i) {
                if (k + i < nk) off[k + i] = run;
                run += v[i];
            }
        }
        carry += all;
        __syncthreads();                                           // (wsum is rewritten by the next tile)
    }
    if (t == 0) {
        off[nk] = carry;
        if (tile_off) coarse[nbins] = carry;
    }
    if (!tile_off) return;
    __syncthreads();
    // tiles per coarse bin and their exclusive scan (nbins <= 512: two bins per work-item)
    u32 tcarry = 0;
    for (u32 b0 = 0; b0 < nbins; b0 += SCAN_ONE_THREADS) {
        const u32 r = b0 + t;
        const u32 x = r < nbins ? (coarse[r + 1] - coarse[r] + MSM_FINE_TILE - 1) / MSM_FINE_TILE : 0;
        const u32 incl = wave_inclusive_scan(x);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 before = tcarry, all = 0;
        ZK_UNROLL for (u32 w = 0; w < NW; ++w) {
            const u32 y = wsum[w];
            if (w < wave) before += y;
            all += y;
        }
        if (r < nbins) tile_off[r] = before + incl - x;
        tcarry += all;
        __syncthreads();
    }
    if (t == 0) tile_off[nbins] = tcarry;
}
// ---- 2a. exclusive scan of the counters: one workgroup per chunk of SCAN_CHUNK counters ----
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_PER_THREAD = 16;
static constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;

static __global__ void k_scan_local(const u32* __restrict__ cnt, u32* __restrict__ off, u32* __restrict__ chunk_sum, u64 total) {
    ZK_PRIO_HIGH();
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
    ZK_PRIO_HIGH();
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
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) off[i] += chunk_sum[i / SCAN_CHUNK];
    if (i == total) off[total] = *grand_total;   // sentinel: off has total+1 entries
}

// ---- 3a. slices of the sorted list ----
// The list (length total = off[nkeys], known only on the device) is cut into nlanes slices of P = max(ceil(total / nlanes),
// min_slice) entries: with nlanes = the number of work-items the machine holds, every work-item of the accumulation
// kernel does the same number of additions in ONE round of workgroups, whatever the scalars look like.
struct MsmCut { u32 nlanes, min_slice; u32 table_len; u32 prio; };   // table_len: entries of a base table (levels x points), for the checked build;
                                                                      // prio: the accumulation's waves raise their issue priority (a lone proof's G2 lane: core.cuh lone_sched)
static __device__ __forceinline__ u32 msm_slice_len(const u32* __restrict__ off, u32 nkeys, MsmCut cut) {
    const u32 total = off[nkeys];
    const u32 P = (total + cut.nlanes - 1) / cut.nlanes;
    return P > cut.min_slice ? P : cut.min_slice;
}
// first key of every lane's slice (upper bound over the offsets)
static __global__ void k_msm_lane_keys(const u32* __restrict__ off, u32 nkeys, MsmCut cut, u32* __restrict__ lane_key, u32* __restrict__ heavy_count) {
    ZK_PRIO_HIGH();
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) *heavy_count = 0;            // (k_msm_find_heavy, the next launch of the stream, counts into it)
    if (g >= cut.nlanes) return;
    const u32 P = msm_slice_len(off, nkeys, cut);
    const u64 pos = (u64)g * P;
    if (pos >= off[nkeys]) { lane_key[g] = nkeys; return; }
    u32 lo = 0, hi = nkeys;          // invariant: off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    lane_key[g] = lo;
}
// buckets whose entries span more than MSM_HEAVY lanes
static __global__ void k_msm_find_heavy(const u32* __restrict__ off, u32 nkeys, MsmCut cut, u32* __restrict__ heavy_list,
                                        u32* __restrict__ heavy_count) {
    ZK_PRIO_HIGH();
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkeys) return;
    const u32 P = msm_slice_len(off, nkeys, cut);
    const u32 b = off[k], e = off[k + 1];
    if (e > b && (e - 1) / P - b / P + 1 > MSM_HEAVY) heavy_list[atomicAdd(heavy_count, 1u)] = k;
}

// ---- 3b. balanced bucket accumulation ----
// Lane g owns sorted entries [g*P, (g+1)*P).  Whenever the walk crosses into another bucket the running sum is
// written to partial[key + g]: along the sorted list (lane, key) only ever increase, so key + lane is unique,
// and bucket `key` finds its partials at the contiguous slots key + g for the lanes g its range overlaps.
// An entry names a table slot (level * stride + point index); the base is fetched packed (one 64-byte line for BN254 G1)
// one entry ahead and unpacked into 29/28-bit limbs when it is used.  A lane that walked no entry of a bucket's tail it was
// assigned (its sum cancelled, or only infinite bases) still writes the slot: the fold reads every slot of a bucket's lanes.
// an empty statement the compiler must have every 32-bit word of `obj` in a register for: values computed before it stay before it
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ void zk_pin_words(T& obj) {
    u32* w = (u32*)&obj;
    ZK_UNROLL for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("" : "+v"(w[i]));
}
#define ZK_PIN_WORDS(x) zk_pin_words(x)
#define ZK_LDS_REREAD() asm volatile("" ::: "memory")
#define ZK_OPAQUE(x) asm volatile("" : "+v"(x))      /* the compiler may not assume it knows this 32-bit value: nothing derived from it is hoisted */
#else
#define ZK_PIN_WORDS(x) ((void)0)
#define ZK_LDS_REREAD() ((void)0)
#define ZK_OPAQUE(x) ((void)0)
#endif
// a point the instruction scheduler may not move anything across: between the products of the hot addition it keeps the compiler
// from interleaving independent products (whose operands and columns would then all be live at once)
#ifndef ZK_ACCUM_FENCE
#define ZK_ACCUM_FENCE 0
#endif
// X of the G1 running sum without its weak reduction (values < 10p, subtractions against 16p): 45 instructions fewer on paper, 34
// scratch accesses more in the compiled loop (2 424 instructions with them against 2 427 without: tools/isa_mix.py g1
// -DZK_ACCUM_X_UNREDUCED=1) — off
#ifndef ZK_ACCUM_X_UNREDUCED
#define ZK_ACCUM_X_UNREDUCED 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_SCHED_FENCE(on) do { if (on) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ZK_SCHED_FENCE(on) ((void)0)
#endif
// SKIP_INF: how a base at infinity is met.  true: the lane sits the step out (one more per-lane branch per step: ≈ 1.5 % more
// instructions on a table without such bases); false: the wavefront's vote sends the step through the general code (free when it
// never happens, twice the price of a step when it does).  Real circuits leave many points at infinity in b_query (every variable
// that does not occur in B: a third of the Poseidon chain's, half of a GM17 key's) — the host picks per table from a count made
// at key load (zkhip_pk::inf_many).
template <class F, int WPE, bool SKIP_INF>
__global__ void __launch_bounds__(256, WPE) k_msm_accum(MsmTables tables, const u32* __restrict__ off, const u32* __restrict__ sorted,
                                                    const u32* __restrict__ lane_key, Xyzz<F>* __restrict__ partial, u64 partial_stride, u32 nkeys, MsmCut cut) {
    constexpr int NW2 = 2 * AffPacked<F>::NW;
#ifndef ZK_EMU
    if (cut.prio) __builtin_amdgcn_s_setprio(2);       // (wave-uniform: a kernel argument)
#endif
    const AffPacked<F>* __restrict__ bases = (const AffPacked<F>*)tables.p[blockIdx.y];   // blockIdx.y: which MSM of the launch
    partial += (u64)blockIdx.y * partial_stride;
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= cut.nlanes) return;
    u32 cur = lane_key[g];
    if (cur >= nkeys) return;
    const u32 total = off[nkeys];
    const u32 P = msm_slice_len(off, nkeys, cut);
    // (lane_key[g] < nkeys says g * P < total, and the list is shorter than 2^32 entries: positions are 32-bit)
    const u32 p0 = g * P;
    const u32 p1 = total - p0 > P ? p0 + P : total;
    // `end`: where the lane's current bucket ends; `nend`: where the NEXT one does — fetched one bucket ahead, so that crossing a
    // boundary never waits for memory (buckets are a few hundred entries deep: some lane of a wavefront crosses one every few
    // steps, and a wait parks all 64 — 8-12 % of the wave cycles of round 4's kernel, profiles/r5a_stall_r4.md)
    u32 end = off[cur + 1], nend = off[cur + 2 < nkeys ? cur + 2 : nkeys];
    ZK_ASSERT_IDX((u64)g * P < total && off[cur] <= p0 && p0 < end && (u64)nkeys + cut.nlanes <= partial_stride);
    // The running sum: X and Y in registers; ZZ and ZZZ in registers too, or (ZZ_IN_LDS) in the lane's own words of LDS — each is
    // read twice and written once per addition, and the 36 registers they would hold through the whole step are what decides
    // whether a second G2 wave fits a SIMD.
    constexpr bool ZZ_LDS = MsmTuning<F>::ZZ_IN_LDS, XY_LDS = MsmTuning<F>::XY_IN_LDS;
    constexpr int FW = (int)(sizeof(F) / 4);
    ZK_DYN_SMEM(acc_smem);                    // msm_accum_lds_bytes<F>(): coordinate `which` (0 ZZ, 1 ZZZ, 2 X, 3 Y; only those that live here), word-major
    u32* const acc_lds = (u32*)acc_smem;
    constexpr int XY_BASE = ZZ_LDS ? 2 : 0;
    F rxy[2] = {F::zero(), F::zero()}, rzz[2] = {F::zero(), F::zero()};
    auto lds_get = [&](int slot) -> F {
        F r;
        u32* wv = (u32*)&r;
        ZK_LDS_REREAD();          // (every fetch is a fresh read: a value kept in registers between its two uses is what this avoids)
        ZK_UNROLL for (int q = 0; q < FW; ++q) wv[q] = acc_lds[(slot * FW + q) * 256 + threadIdx.x];
        return r;
    };
    auto lds_put = [&](int slot, const F& v) {
        const u32* wv = (const u32*)&v;
        ZK_UNROLL for (int q = 0; q < FW; ++q) acc_lds[(slot * FW + q) * 256 + threadIdx.x] = wv[q];
    };
    auto get_zz = [&](int which) -> F { return ZZ_LDS ? lds_get(which) : rzz[which]; };
    auto put_zz = [&](int which, const F& v) { if (ZZ_LDS) lds_put(which, v); else rzz[which] = v; };
    auto get_xy = [&](int which) -> F { return XY_LDS ? lds_get(XY_BASE + which) : rxy[which]; };
    auto put_xy = [&](int which, const F& v) { if (XY_LDS) lds_put(XY_BASE + which, v); else rxy[which] = v; };
    auto whole = [&]() -> Xyzz<F> { return {get_xy(0), get_xy(1), get_zz(0), get_zz(1)}; };
    auto set_whole = [&](const Xyzz<F>& t) { put_xy(0, t.x); put_xy(1, t.y); put_zz(0, t.zz); put_zz(1, t.zzz); };
    set_whole(Xyzz<F>::inf());
    // the sorted entries travel TWO steps ahead of their use (e: the next step's, whose base is fetched now; e_ahead: the one
    // after), the bases one: no address of this loop is computed from a load of the same step
    u32 e = sorted[p0], e_ahead = sorted[p0 + 1 < p1 ? p0 + 1 : p0];
    ZK_ASSERT_IDX((e & 0x7fffffffu) < cut.table_len);
    constexpr bool IN_REGS = MsmTuning<F>::PREFETCH_REGS;
    u32 w[NW2];
    u32 touched = 0;
    if (IN_REGS) aff_load_words<F>(bases, e & 0x7fffffffu, w);
    // `first`: the next point STARTS a sum (slice start, bucket boundary, or the sum so far cancelled to infinity)
    bool first = true;
    for (u32 pos = p0; pos < p1; ++pos) {
        // PREFETCH_REGS: the packed words are consumed by the unpacking and the same registers take the NEXT base at once (always:
        // the last iteration fetches its own entry again): the fetch has the whole addition to arrive in, nothing is copied from a
        // "next" set of registers to a "current" one, and no lane skips the load.  Otherwise the base is loaded where it is used
        // and the NEXT one is only touched (one word: the line travels to the cache meanwhile).
        const u32 e_cur = e;
        if (!IN_REGS) aff_load_words<F>(bases, e_cur & 0x7fffffffu, w);
        Aff<F> pt = aff_unpack<F>(w);
        if (IN_REGS) ZK_PIN_WORDS(pt);          // (the unpacking is complete HERE: the fetch below may overwrite the packed words in place)
        const bool neg = (e_cur & 0x80000000u) != 0;
        if (pos == end) {
            const u32 done = cur;
            do {           // (more than one round only across empty buckets; the new fetch goes out BEFORE the stores below: taking
                ++cur;     // `nend` waits for whatever is in flight, and at this point that is nothing recent)
                ZK_ASSERT_IDX(cur < nkeys);
                end = nend;
                nend = off[cur + 2 < nkeys ? cur + 2 : nkeys];
            } while (end <= pos);
            // (the slot's address is computed HERE from the lane's number: kept across the loop as a lane-invariant 64-bit value it
            // was spilled, and its reload waited for the fetch above)
            u32 g_here = g;
            ZK_OPAQUE(g_here);
            partial[(u64)done + g_here] = whole();  // (invariant: whenever `first` holds here, the sum IS the empty one — see the general path)
            first = true;
        }
        // (the fetches of this step are issued AFTER the boundary code: what that code waits for — the bucket end fetched a whole bucket
        // ago, as far as the hardware's in-order counter is concerned everything in flight — must not include loads issued just now)
        e = e_ahead;                                   // entry pos + 1 (the last step fetches its own entry's base again)
        e_ahead = sorted[pos + 2 < p1 ? pos + 2 : p1 - 1];
        ZK_ASSERT_IDX((e & 0x7fffffffu) < cut.table_len);
        if (IN_REGS) aff_load_words<F>(bases, e & 0x7fffffffu, w);
        else touched |= ((const volatile u32*)(bases + (e & 0x7fffffffu)))[0];
        // Lanes of one wavefront are at different places of their slices: at every step some lane starts a new bucket while the
        // others add.  Written as nested per-lane branches (empty sum? equal x? infinite base?) that costs three re-convergence
        // points per step with a copy of the whole accumulator at each.  Instead EVERY lane computes Pp and R, and the one case
        // the branch-free addition cannot take — Pp = 0: doubling or cancellation — is looked for across the wavefront first: if
        // ANY lane has it, the whole wavefront runs the general code for this step (a uniform branch, taken practically never
        // on full-width scalars).  Otherwise a lane whose base is the point at infinity sits the step out (real circuits leave
        // many of those in b_query: every variable that does not occur in B), a lane that starts a sum takes the point itself,
        // and the others add.
        const bool pinf = pt.is_inf();
        const F ys = fe_cneg(pt.y, neg);
        // (base field: X1 is kept as the numerator of X3 — TIGHT, value < 10p — without the weak reduction: it only ever meets a
        // product (Q = X1 PP) or a subtraction with a large enough multiple of p; bounds below.  Fq2 keeps the reduction: the
        // square of Pp in its (a0 + a1)(a0 - a1) form has no room for components of 18p.)
        constexpr bool XWIDE = ZK_ACCUM_X_UNREDUCED && !MsmTuning<F>::IS_EXT;
        constexpr int KX = XWIDE ? 16 : 4;
        const F Pp = fe_sub_k<KX>(ecm_k<true>(get_zz(0), pt.x), get_xy(0));    // X1 < 3p (10p);  Pp < 6p (18p: PP < 3p, PPP, Q, ZZ3 < 2p all the same)
        const F R = fe_sub_k<4>(ecm_k<true>(get_zz(1), ys), get_xy(1));        // Y1 < 4p;  R < 6p
        const bool special = SKIP_INF ? (!pinf && !first && fe_is_zero_modp(Pp)) : (pinf || (!first && fe_is_zero_modp(Pp)));
        if (ZK_WAVE_ANY(special)) {
            // (the base is fetched AGAIN here: holding its coordinates across the vote for a path that practically never runs costs
            // the hot path their registers — 36 for G2, which the compiler then spills on every step)
            u32 wg[NW2];
            aff_load_words<F>(bases, e_cur & 0x7fffffffu, wg);
            const Aff<F> pg = aff_unpack<F>(wg);
            Xyzz<F> t = first ? Xyzz<F>::inf() : whole();
            if (XWIDE) t.x = fe_relax(t.x);         // (the general addition expects X1 < 4p)
            if (!pinf) {
                if (ZK_ACCUM_COLD_CALL) t = xyzz_madd_cold(t, Aff<F>{pg.x, fe_cneg(pg.y, neg)});
                else xyzz_madd_acc<true>(t, Aff<F>{pg.x, fe_cneg(pg.y, neg)});
            }
            set_whole(t);
            first = t.is_inf();
        } else if (SKIP_INF && pinf) {
            if (first) set_whole(Xyzz<F>::inf());      // (keeps the invariant: `first` and a stale sum never meet a store)
        } else if (first) {
            put_xy(0, pt.x); put_xy(1, ys); put_zz(0, F::one()); put_zz(1, F::one());
            first = false;
        } else {
            // the addition proper (ec.cuh xyzz_madd_finish, with ZZ / ZZZ fetched where they are multiplied)
            constexpr bool FENCE = (ZK_ACCUM_FENCE & (MsmTuning<F>::IS_EXT ? 2 : 1)) != 0;
            ZK_SCHED_FENCE(FENCE);
            const F PP = ecs<true>(Pp);
            ZK_SCHED_FENCE(FENCE);
            put_zz(0, ecm_k<true>(get_zz(0), PP));
            ZK_SCHED_FENCE(FENCE);
            const F PPP = ecm<true>(Pp, PP);
            ZK_SCHED_FENCE(FENCE);
            put_zz(1, ecm_k<true>(get_zz(1), PPP));
            ZK_SCHED_FENCE(FENCE);
            const F Q = ecm_k<true>(get_xy(0), PP);
            ZK_SCHED_FENCE(FENCE);
            const F X3n = fu_x3_numerator(ecs<true>(R), PPP, Q);              // R^2 - PPP - 2Q: TIGHT, < 10p
            const F X3 = XWIDE ? X3n : fe_relax(X3n);                          // (< 3p after the weak reduction)
            ZK_SCHED_FENCE(FENCE);
            put_xy(1, ec_mulsub<true>(R, fe_sub_k<KX>(Q, X3), get_xy(1), PPP));  // R < 6p times < 6p (18p) + Y1 (< 2p) PPP: one reduction, < 2p (Fq2: < 3p)
            put_xy(0, X3);
            ZK_SCHED_FENCE(FENCE);
        }
    }
    partial[(u64)cur + g] = whole();
    (void)touched;      // (the volatile loads stay)
}

// sum of one bucket's partials (after the heavy pass the first slot of a heavy bucket holds its total)
template <class F>
__device__ __forceinline__ Xyzz<F> msm_bucket_sum(const Xyzz<F>* partial, const u32* __restrict__ off, u32 key, u32 P) {
    const u32 b = off[key], e = off[key + 1];
    ZK_ASSERT_IDX(b <= e);
    if (e <= b) return Xyzz<F>::inf();
    const u32 g0 = b / P, g1 = (e - 1) / P;
    Xyzz<F> s = partial[(u64)key + g0];
    if (g1 - g0 + 1 <= MSM_HEAVY)
        for (u32 g = g0 + 1; g <= g1; ++g) xyzz_add_from(s, &partial[(u64)key + g]);
    return s;
}

// workgroup tree sum over sh[0 .. blockDim.x) (blockDim.x a power of two); result in sh[0]
// (a level reads slots >= st and writes slots < st: one barrier per level)
template <class F>
__device__ __forceinline__ void block_tree_sum(Xyzz<F>* sh) {
    __syncthreads();
    for (unsigned st = blockDim.x >> 1; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            Xyzz<F> a = sh[threadIdx.x];
            xyzz_add_from(a, &sh[threadIdx.x + st]);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
}
// the same with the work-item's value `v` kept in registers from level to level (v of work-item 0 ends as the sum; sh[t] = v on entry)
template <class F>
__device__ __forceinline__ void block_tree_sum_reg(Xyzz<F>* sh, Xyzz<F>& v, unsigned width) {
    __syncthreads();
    for (unsigned st = width >> 1; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            xyzz_add_from(v, &sh[threadIdx.x + st]);
            if (st > 1) sh[threadIdx.x] = v;
        }
        __syncthreads();
    }
}

// k * p for a small unsigned k (left-to-right double-and-add over the significant bits only)
template <class F>
__device__ __forceinline__ Xyzz<F> xyzz_mul_small(const Xyzz<F>& p, u32 k) {
    Xyzz<F> r = Xyzz<F>::inf();
    if (k == 0 || p.is_inf()) return r;
    for (int i = 31 - __clz(k); i >= 0; --i) {
        r = xyzz_dbl_inl(r);
        if ((k >> i) & 1) xyzz_add_acc(r, p);
    }
    return r;
}

// ---- 5. window fold: sum_{b < K} (b+1) * B_b, two-digit form ----
// Bucket index b = hi * Lw + lo (Lw = min(K, 256) buckets per row, H = K / Lw rows):
//     sum (b+1) B_b = sum_lo (lo+1) * C_lo  +  Lw * sum_hi hi * R_hi,      R_hi = row sums, C_lo = column sums.
// Both digit sums are plain (unweighted) reductions over all buckets — wide and shallow — and only the Lw + H
// row/column totals per window need a (short) double-and-add.
//
// 5a. one workgroup per (row, bucket set): combine each bucket's partials, keep the bucket value for the column pass,
//     tree-sum the row.  A heavy bucket (more than MSM_HEAVY partials; k_msm_find_heavy lists them) is first summed by
//     the whole workgroup of its row into its first slot — no kernel of its own: a launch that only finds an empty list
//     still waits for a place on a machine full of accumulation waves (5 ms in a trace).
//     A VERY heavy bucket — a witness of bits puts every variable that is 1 into bucket 1 of the lowest window: a quarter of a
//     SHA-256 circuit's entries, tens of thousands of partials — would keep that one workgroup busy for milliseconds (2.7 ms for
//     G2 in a trace of zokrates_amd/sha256_circuit.py), so k_msm_heavy_reduce goes first: the partials of a heavy bucket are cut
//     into MSM_HEAVY_CHUNKS runs, workgroup c sums run c of every heavy bucket into the run's first slot, and the row's
//     workgroup below only adds the run heads.  With an empty list the kernel returns at once.
static constexpr u32 MSM_HEAVY_CHUNKS = 64;
struct HeavyRun {
    u32 g0, g1, len;     // the bucket's lanes [g0, g1], lanes per run
};
static __device__ __forceinline__ HeavyRun msm_heavy_run(const u32* __restrict__ off, u32 key, u32 P) {
    HeavyRun r;
    r.g0 = off[key] / P;
    r.g1 = (off[key + 1] - 1) / P;
    r.len = (r.g1 - r.g0 + MSM_HEAVY_CHUNKS) / MSM_HEAVY_CHUNKS;
    return r;
}
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_heavy_reduce(Xyzz<F>* partial, u64 partial_stride, const u32* __restrict__ off, u32 nkeys, MsmCut cut,
                                                                                  const u32* __restrict__ heavy_list, const u32* __restrict__ heavy_count) {
    ZK_PRIO_HIGH();
    const u32 nh = *heavy_count;
    if (nh == 0) return;
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    partial += (u64)blockIdx.y * partial_stride;           // blockIdx.y: which MSM of the launch
    const u32 P = msm_slice_len(off, nkeys, cut), lo = threadIdx.x;
    for (u32 h = 0; h < nh; ++h) {
        const u32 hk = heavy_list[h];
        const HeavyRun r = msm_heavy_run(off, hk, P);
        const u32 b = r.g0 + blockIdx.x * r.len;
        if (b > r.g1 || r.len == 1) continue;    // (uniform over the workgroup)
        const u32 e = b + r.len - 1 < r.g1 ? b + r.len - 1 : r.g1;
        Xyzz<F> s = Xyzz<F>::inf();
        for (u32 g = b + lo; g <= e; g += blockDim.x) xyzz_add_from(s, &partial[(u64)hk + g]);
        sh[lo] = s;
        block_tree_sum<F>(sh);
        if (lo == 0) partial[(u64)hk + b] = sh[0];
        __syncthreads();
    }
}
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_fold_rows(Xyzz<F>* partial, u64 partial_stride, const u32* __restrict__ off, u32 nkeys, MsmCut cut, u32 K, u32 Lw,
                                                        const u32* __restrict__ heavy_list, const u32* __restrict__ heavy_count, u32 heavy_runs,
                                                        Xyzz<F>* __restrict__ bucket, Xyzz<F>* __restrict__ rows) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    partial += (u64)blockIdx.z * partial_stride;           // blockIdx.z: which MSM of the launch
    bucket += (u64)blockIdx.z * nkeys;
    rows += (u64)blockIdx.z * gridDim.y * gridDim.x;
    const u32 j = blockIdx.y, hi = blockIdx.x, lo = threadIdx.x;
    const u32 P = msm_slice_len(off, nkeys, cut);
    const u32 row0 = j * K + hi * Lw;
    const u32 nh = *heavy_count;
    for (u32 h = 0; h < nh; ++h) {
        const u32 hk = heavy_list[h];
        if (hk - row0 >= Lw) continue;           // (uniform over the workgroup)
        const HeavyRun r = msm_heavy_run(off, hk, P);
        const u32 g0 = r.g0, g1 = r.g1, step = heavy_runs ? r.len : 1;     // heavy_runs: k_msm_heavy_reduce left one sum per run
        Xyzz<F> s = Xyzz<F>::inf();
        for (u64 g = g0 + (u64)lo * step; g <= g1; g += (u64)blockDim.x * step) xyzz_add_from(s, &partial[(u64)hk + g]);
        sh[lo] = s;
        block_tree_sum<F>(sh);
        if (lo == 0) partial[(u64)hk + g0] = sh[0];
        __syncthreads();
    }
    const u32 key = row0 + lo;
    ZK_ASSERT_IDX(key < nkeys && (u64)key + (off[key + 1] ? (off[key + 1] - 1) / P : 0) < partial_stride);
    Xyzz<F> v = msm_bucket_sum<F>(partial, off, key, P);
    bucket[key] = v;
    sh[lo] = v;
    block_tree_sum<F>(sh);
    if (lo == 0) rows[(u64)j * gridDim.x + hi] = sh[0];
}
// 5a'. rows AND columns in one launch (H, Lw <= 256: every resident key).  The column sums do not need the row pass: a workgroup
//     per LINE of the H x Lw bucket matrix — blockIdx.y = 2 set + dir; dir 0: row blockIdx.x, one work-item per lo; dir 1: column
//     blockIdx.x, one work-item per hi — combines its buckets' partials and tree-sums them; the bucket values are computed twice
//     (a few additions per bucket, against the ~240 that made it) and never stored.  One kernel and 3-5 + 8 dependent additions
//     where round 5 ran two (rows, then columns over the stored bucket values: 8 serial + 5 tree levels more) — the fold is a chain
//     of dependent additions at the end of every MSM, and at the end of a lone proof nothing hides it.
//     Heavy buckets: the whole workgroup sums the bucket's partials (or the run heads k_msm_heavy_reduce left) and hands the total
//     to the bucket's work-item through LDS — nothing is written back, the other direction's workgroup reads the same slots.
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_fold_lines(const Xyzz<F>* __restrict__ partial, u64 partial_stride, const u32* __restrict__ off, u32 nkeys,
                                                                                MsmCut cut, u32 K, u32 Lw, u32 H, const u32* __restrict__ heavy_list,
                                                                                const u32* __restrict__ heavy_count, u32 heavy_runs, Xyzz<F>* __restrict__ rows,
                                                                                Xyzz<F>* __restrict__ cols) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 sets = gridDim.y >> 1, j = blockIdx.y >> 1, dir = blockIdx.y & 1u, line = blockIdx.x, t = threadIdx.x;
    const u32 len = dir ? H : Lw, nlines = dir ? Lw : H;       // work-items of this line, lines of this direction
    if (line >= nlines) return;                                // (uniform: the grid is max(H, Lw) wide)
    partial += (u64)blockIdx.z * partial_stride;               // blockIdx.z: which MSM of the launch
    const u32 P = msm_slice_len(off, nkeys, cut);
    const u32 set0 = j * K;
    const bool live = t < len;
    const u32 key = set0 + (dir ? t * Lw + line : line * Lw + t);
    Xyzz<F> v = Xyzz<F>::inf();
    bool have = !live;
    const u32 nh = *heavy_count;
    for (u32 h = 0; h < nh; ++h) {
        const u32 hk = heavy_list[h];
        const u32 rel = hk - set0;                             // (unsigned: a key of another set wraps far beyond K)
        if (rel >= K || (dir ? rel % Lw : rel / Lw) != line) continue;     // (uniform over the workgroup)
        const HeavyRun r = msm_heavy_run(off, hk, P);
        const u32 step = heavy_runs ? r.len : 1;               // heavy_runs: k_msm_heavy_reduce left one sum per run, at the run's first slot
        Xyzz<F> s = Xyzz<F>::inf();
        for (u64 g = r.g0 + (u64)t * step; g <= r.g1; g += (u64)blockDim.x * step) xyzz_add_from(s, &partial[(u64)hk + g]);
        sh[t] = s;
        block_tree_sum_reg<F>(sh, s, blockDim.x);
        if (t == 0) sh[0] = s;
        __syncthreads();
        if (live && key == hk) { v = sh[0]; have = true; }
        __syncthreads();
    }
    if (!have) {
        ZK_ASSERT_IDX(key < nkeys && (u64)key + (off[key + 1] ? (off[key + 1] - 1) / P : 0) < partial_stride);
        v = msm_bucket_sum<F>(partial, off, key, P);
    }
    sh[t] = v;
    block_tree_sum_reg<F>(sh, v, blockDim.x);
    if (t == 0) (dir ? cols : rows)[((u64)blockIdx.z * sets + j) * nlines + line] = v;
}
// 5b. column sums: work-item (lo, hg) adds its share of the rows serially, then the HG shares are tree-added.
//     blockDim = (CW, HG); grid = (Lw / CW, sets * NG, MSMs of the launch).  The H rows of a set are cut into NG = H / RG groups
//     of RG rows (blockIdx.y = set * NG + group) and every group leaves its own column sums: with few rows NG = 1 and the
//     result is final; with thousands of rows (wide windows) a second launch of this kernel adds the NG partial results — the
//     serial part of either stays at RG / HG additions.  src: element (set j, row hi, column lo) of MSM z at
//     src[z * msm_stride + j * set_stride + hi * Lw + lo]; dst[((z * sets + j) * NG + g) * Lw + lo].
template <class F>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_fold_cols(const Xyzz<F>* __restrict__ src, u64 set_stride, u64 msm_stride, u32 Lw, u32 H, u32 RG,
                                                                               Xyzz<F>* __restrict__ dst) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 NG = H / RG, sets = gridDim.y / NG;
    const u32 j = blockIdx.y / NG, g = blockIdx.y % NG;
    const u32 lo = blockIdx.x * blockDim.x + threadIdx.x, hg = threadIdx.y, HG = blockDim.y;
    src += (u64)blockIdx.z * msm_stride + (u64)j * set_stride + (u64)g * RG * Lw;
    dst += (((u64)blockIdx.z * sets + j) * NG + g) * Lw;
    Xyzz<F> s = Xyzz<F>::inf();
    for (u32 hi = hg; hi < RG; hi += HG) xyzz_add_from(s, &src[(u64)hi * Lw + lo]);
    sh[hg * blockDim.x + threadIdx.x] = s;
    __syncthreads();
    for (unsigned st = HG >> 1; st > 0; st >>= 1) {
        if (hg < st) {
            Xyzz<F> a = sh[hg * blockDim.x + threadIdx.x];
            xyzz_add_from(a, &sh[(hg + st) * blockDim.x + threadIdx.x]);
            sh[hg * blockDim.x + threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (hg == 0) dst[lo] = sh[threadIdx.x];
}
// 5c. one workgroup per bucket set: the two weighted digit sums.
template <class F, class FS>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_fold_final(const Xyzz<F>* __restrict__ rows, const Xyzz<F>* __restrict__ cols, u32 Lw, u32 H,
                                                         Xyzz<FS>* __restrict__ window_sum, u32 sum_stride) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 j = blockIdx.x, t = threadIdx.x;
    rows += (u64)blockIdx.y * gridDim.x * H;               // blockIdx.y: which MSM of the launch
    cols += (u64)blockIdx.y * gridDim.x * Lw;
    window_sum += (u64)blockIdx.y * sum_stride;
    // sum_hi hi * R_hi, then times Lw (a power of two: log2 doublings)
    Xyzz<F> term = Xyzz<F>::inf();
    for (u32 hi = t; hi < H; hi += blockDim.x) xyzz_add_acc(term, xyzz_mul_small(rows[(u64)j * H + hi], hi));
    sh[t] = term;
    block_tree_sum<F>(sh);
    Xyzz<F> hi_sum = sh[0];
    __syncthreads();
    // sum_lo (lo + 1) * C_lo
    term = Xyzz<F>::inf();
    for (u32 lo = t; lo < Lw; lo += blockDim.x) xyzz_add_acc(term, xyzz_mul_small(cols[(u64)j * Lw + lo], lo + 1));
    sh[t] = term;
    block_tree_sum<F>(sh);
    if (t == 0) {
        for (u32 q = Lw; q > 1; q >>= 1) hi_sum = xyzz_dbl_inl(hi_sum);
        Xyzz<F> r = sh[0];
        xyzz_add_acc(r, hi_sum);
        window_sum[2 * j] = xyzz_to_sat<FS>(r);         // (a bucket set's value is the sum of its two entries: see 5c')
        window_sum[2 * j + 1] = Xyzz<FS>::inf();
    }
}

// 5c'. the same result with a shorter dependent chain (the fold tail is pure latency: one workgroup per digit and set).
// A weighted sum of n totals is a sum of suffix sums — sum_k (k+1) V_k = sum_k S_k with S_k = sum_{i >= k} V_i, and
// sum_k k V_k = sum_{k >= 1} S_k — so a digit costs one parallel suffix scan plus one tree sum (2 log2 n full additions)
// instead of a double-and-add per element.  Every digit of the bucket index is its own workgroup (blockIdx.z) of <= 256
// work-items — one wave per SIMD, the whole register file, no scratch even for G2 — and leaves its own sum, already
// multiplied by the power of two its digit weighs (`dbl` doublings): window_sum[nd j + d]; the host adds the nd sums
// (msm_combine).  Two digits: column and row of the bucket index.
struct FoldDigit {
    const void* src;     // totals of this digit: element t of (MSM z, set j) at src[((z * sets + j) * len + t]
    u32 len;             // a power of two <= 256
    u32 plus_one;        // weights k + 1 (the digit that carries the "+1" of bucket b <-> digit value b + 1), else k
    u32 dbl;             // log2 of the digit's place value
};
struct FoldDigits { FoldDigit d[2]; };
template <class F, class FS>
__global__ void __launch_bounds__(256, MsmTuning<F>::COLD_WPE) k_msm_fold_final_scan(FoldDigits digs, Xyzz<FS>* __restrict__ window_sum, u32 sum_stride) {
    ZK_PRIO_HIGH();
    ZK_DYN_SMEM(smem);
    Xyzz<F>* sh = (Xyzz<F>*)smem;
    const u32 j = blockIdx.x, t = threadIdx.x, nd = gridDim.z;
    const FoldDigit dg = blockIdx.z == 0 ? digs.d[0] : digs.d[1];
    const u32 seglen = dg.len;
    const Xyzz<F>* src = (const Xyzz<F>*)dg.src + ((u64)blockIdx.y * gridDim.x + j) * seglen;     // blockIdx.y: which MSM of the launch
    window_sum += (u64)blockIdx.y * sum_stride;
    const bool live = t < seglen;
    Xyzz<F> v = Xyzz<F>::inf();
    if (live) v = src[t];
    sh[t] = v;
    __syncthreads();
    for (u32 d = 1; d < seglen; d <<= 1) {              // suffix scan: a step reads (the partner's value, from LDS, inside the addition), then writes
        const bool has = live && t + d < seglen;
        if (has) xyzz_add_from(v, &sh[t + d]);
        __syncthreads();
        if (has) sh[t] = v;
        __syncthreads();
    }
    if (!dg.plus_one && t == 0) {                       // S_0 carries weight 0
        v = Xyzz<F>::inf();
        sh[0] = v;
    }
    block_tree_sum_reg<F>(sh, v, seglen);
    if (t == 0) {
        Xyzz<F> r = v;
        for (u32 q = 0; q < dg.dbl; ++q) r = xyzz_dbl_inl(r);
        window_sum[nd * j + blockIdx.z] = xyzz_to_sat<FS>(r);
    }
}

// ---- precomputed window multiples (key load) ----
// tbl[j * stride + i] = 2^(c j) * tbl[i] for j = 1 .. W-1 and i in [i0, i0 + cnt) (c: bits between two levels, W: levels): one work-item per base walks the
// doublings (XYZZ, kept in `tmp`), then turns its W-1 points back into affine form with ONE field inversion (Montgomery's
// trick over its own ZZZ values, prefix products in `pre`) and packs them.  tmp / pre: (W-1) x cnt entries, level-major.
template <class F, class FS>
__global__ void __launch_bounds__(256) k_msm_table_levels(AffPacked<F>* __restrict__ tbl, u64 stride, u64 i0, u64 cnt, int c, int W,
                                                           Xyzz<F>* __restrict__ tmp, F* __restrict__ pre) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    u32 w[2 * AffPacked<F>::NW];
    aff_load_words<F>(tbl, i0 + i, w);
    Xyzz<F> P = Xyzz<F>::from_affine(aff_unpack<F>(w));
    F acc = F::one();
    for (int j = 1; j < W; ++j) {
        for (int b = 0; b < c; ++b) P = xyzz_dbl_inl(P);
        tmp[(u64)(j - 1) * cnt + i] = P;
        if (!P.is_inf()) {
            pre[(u64)(j - 1) * cnt + i] = acc;
            acc = ec_mul(acc, P.zzz);
        }
    }
    F inv = ec_inv(acc);                         // acc is a product of non-zero ZZZ values
    for (int j = W - 1; j >= 1; --j) {
        const Xyzz<F> Q = tmp[(u64)(j - 1) * cnt + i];
        Aff<F> a = Aff<F>::inf();
        if (!Q.is_inf()) {
            const F i3 = ec_mul(inv, pre[(u64)(j - 1) * cnt + i]);   // 1 / ZZZ_j
            inv = ec_mul(inv, Q.zzz);
            const F i2 = ec_sqr(ec_mul(Q.zz, i3));                   // (ZZ / ZZZ)^2 = 1 / ZZ
            a.x = ec_mul(Q.x, i2);
            a.y = ec_mul(Q.y, i3);
        }
        aff_pack(a, tbl + (u64)j * stride + i0 + i);
    }
}

// ---- key preparation ----
// bitmap[i / 32] bit i % 32 |= "point i of this table is NOT the point at infinity" (the caller starts from all zero and runs the
// tables of a family through it: a variable stays in the family's sort if any of its bases is finite)
template <class F>
__global__ void k_mark_finite(const AffPacked<F>* __restrict__ tbl, u64 n, u32* __restrict__ bitmap) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[2 * AffPacked<F>::NW];
    aff_load_words<F>(tbl, i, w);
    u32 any = 0;
    ZK_UNROLL for (int q = 0; q < 2 * AffPacked<F>::NW; ++q) any |= w[q];
    if (any) atomicOr(&bitmap[i >> 5], 1u << (i & 31));
}
// how many of the n packed points are the point at infinity (all-zero)
template <class F>
__global__ void k_count_infinite(const AffPacked<F>* __restrict__ tbl, u64 n, u32* __restrict__ count) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[2 * AffPacked<F>::NW];
    aff_load_words<F>(tbl, i, w);
    u32 any = 0;
    ZK_UNROLL for (int q = 0; q < 2 * AffPacked<F>::NW; ++q) any |= w[q];
    if (!any) atomicAdd(count, 1u);
}
// out[p] = (idx < n_src) ? in[idx] : infinity, idx = natural index of sigma position p (h_query layout)
template <class PT>
__global__ void k_sigma_gather_points(const PT* __restrict__ in, PT* __restrict__ out, u64 n, u64 n_src, u32 n1, u32 n2, u32 n3) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    u64 nat = sigma_nat(p, n1, n2, n3);
    out[p] = nat < n_src ? in[nat] : PT::inf();
}

// fixed-base multiplication for setup (N3): out[i] = k_i * G with tbl[j*256 + d] = d * 2^(8j) * G (affine)
template <class F>
__global__ void __launch_bounds__(64) k_fixed_base_mul(const u32* __restrict__ scalars, u64 n, const Aff<F>* __restrict__ tbl, int nwin,
                                                        Aff<F>* __restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32* k = scalars + i * 8;
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int j = 0; j < nwin; ++j) {
        const u32 d = (k[j >> 2] >> ((j & 3) * 8)) & 0xffu;
        if (d) xyzz_madd_acc(acc, tbl[(size_t)j * 256 + d]);
    }
    out[i] = xyzz_to_affine(acc);
}

// tbl[j*256 + d] = d * P_j (affine), P_j = 2^(8j) G given in pj[]
template <class F>
__global__ void __launch_bounds__(64) k_fixed_base_table(const Aff<F>* __restrict__ pj, Aff<F>* __restrict__ tbl, int nwin) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin * 256) return;
    const int j = t >> 8, d = t & 255;
    tbl[t] = xyzz_to_affine(xyzz_mul_u32(Xyzz<F>::from_affine(pj[j]), (u32)d));
}

}  // namespace zk

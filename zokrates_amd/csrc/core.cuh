// core.cuh — host side of libzkhip.so: contexts, key/constraint-system residency and the Groth16 prover
// schedule, templated on the curve.  Instantiated once per curve in curve_bn254.hip / curve_bls381.hip
// (separate translation units so the two curves compile in parallel); the C ABI lives in zkhip_api.hip.
//
// Replaces `<Ark as Backend<T, G16>>::generate_proof` (/root/reference/zokrates_ark/src/groth16.rs:20-53)
// from the point where the reference hands over to ark: `ProvingKey::deserialize_unchecked` (:40-42) and
// `Groth16::prove` (:44).  SURVEY.md §8(a) rows a7, a8, K1-K9.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/zkhip.h"
#include "devrt.h"
#include "ec.cuh"
#include "kernels_msm.cuh"
#include "kernels_ntt.cuh"

namespace zk {

// ------------------------------------------------------------------ curves
struct CurveBn254 {
    static constexpr int ID = ZKHIP_CURVE_BN128;
    typedef Fe<Bn254Fr> Fr;
    typedef Fe<Bn254Fq> Fq;
    typedef Fe2<Bn254Fq> Fq2;
    static constexpr u64 GENERATOR = 5;       // Fr::GENERATOR, the coset shift g
    static constexpr int TWO_ADICITY = 28;
    // standard generators, canonical limbs (G1: x, y; G2: x.c0, x.c1, y.c0, y.c1); BN254's are the ones pinned by
    // /root/reference/zokrates_proof_systems/src/solidity.rs:430-441
    static const u32* g1_gen() { static const u32 t[] = {0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000002u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}; return t; }
    static const u32* g2_gen() { static const u32 t[] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu, 0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u, 0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u, 0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u}; return t; }
};
struct CurveBls381 {
    static constexpr int ID = ZKHIP_CURVE_BLS12_381;
    typedef Fe<Bls381Fr> Fr;
    typedef Fe<Bls381Fq> Fq;
    typedef Fe2<Bls381Fq> Fq2;
    static constexpr u64 GENERATOR = 7;
    static constexpr int TWO_ADICITY = 32;
    // standard generators, canonical limbs (G1: x, y; G2: x.c0, x.c1, y.c0, y.c1); BN254's are the ones pinned by
    // /root/reference/zokrates_proof_systems/src/solidity.rs:430-441
    static const u32* g1_gen() { static const u32 t[] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu, 0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u, 0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu, 0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u}; return t; }
    static const u32* g2_gen() { static const u32 t[] = {0xc121bdb8u, 0xd48056c8u, 0xa805bbefu, 0x0bac0326u, 0x7ae3d177u, 0xb4510b64u, 0xfa403b02u, 0xc6e47ad4u, 0x2dc51051u, 0x26080527u, 0xf08f0a91u, 0x024aa2b2u, 0x5d042b7eu, 0xe5ac7d05u, 0x13945d57u, 0x334cf112u, 0xdc7f5049u, 0xb5da61bbu, 0x9920b61au, 0x596bd0d0u, 0x88274f65u, 0x7dacd3a0u, 0x52719f60u, 0x13e02b60u, 0x08b82801u, 0xe1935486u, 0x3baca289u, 0x923ac9ccu, 0x5160d12cu, 0x6d429a69u, 0x8cbdd3a7u, 0xadfd9baau, 0xda2e351au, 0x8cc9cdc6u, 0x727d6e11u, 0x0ce5d527u, 0xf05f79beu, 0xaaa9075fu, 0x5cec1da1u, 0x3f370d27u, 0x572e99abu, 0x267492abu, 0x85a763afu, 0xcb3e287eu, 0x2bc28b99u, 0x32acd2b0u, 0x2ea734ccu, 0x0606c4a0u}; return t; }
};

struct ApiError {
    int32_t code;
    std::string msg;
};
static inline void require(bool ok, int32_t code, const char* msg) {
    if (!ok) throw ApiError{code, msg};
}

// ------------------------------------------------------------------ device buffers
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    DBuf() {}
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { dev_free(p); }
    void swap(DBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); }
    void release() { dev_free(p); p = nullptr; cap = 0; }
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        dev_free(p);
        p = nullptr;
        cap = 0;
        p = dev_alloc(bytes);
        cap = bytes;
    }
};
template <class T> static inline T* ptr(const DBuf& b) { return (T*)b.p; }

static inline unsigned blocks_for(u64 n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }
static inline int ilog2_floor(u64 x) { int l = 0; while (x >>= 1) ++l; return l; }
static inline int ilog2_ceil(u64 x) { int l = 0; while (((u64)1 << l) < x) ++l; return l; }

}  // namespace zk

using namespace zk;

// ------------------------------------------------------------------ context
struct NttPlanBase {
    int curve, logN;
    virtual ~NttPlanBase() {}
};
// result of one classification / digit / counting-sort pass; shared (read-only) by every base set paired with those scalars
struct MsmSort {
    DBuf dig, sorted, cnt, off, cursor, chunk_sum, grand;   // dig: the scalars' signed digits, window-major (k_msm_digits)
    DBuf pairs, ccur, tile_off;                             // the two-level placement: (key & 255, entry) pairs by coarse bin, the bins' cursors, tiles per bin
    Event ready = nullptr;   // recorded on the main stream when the pass is complete
};
// workspace and stream of one MSM: the five MSMs of a proof are independent once their scalars are sorted, and the
// fold stages are latency-bound (few, long dependent chains), so they run concurrently and fill each other's gaps
struct MsmLane {
    Stream stream = 0;        // made on first use (lane_stream): a stream is ~10 ms of queue set-up, and a process that proves once on
    bool made = false;        // one stream (the CLI: ZKHIP_TUNE_SERIAL) never needs the ~20 a resident prover keeps busy
    bool high_priority = false;
    DBuf lane_key, heavy, partial, bucket, rows, cols;
    Event done = nullptr;
    Stream fold_stream = 0;   // a second stream for the lane's fold chain (zkhip_ctx::fold_hop; made on first use: lane_fold_stream)
    bool fold_made = false;
    bool fold_owned = true;   // (false: one stream per lane TYPE, shared by the slots and owned by the context — make_pipe_streams)
    Event acc_done = nullptr; // ... which waits for this event behind the lane's accumulation
    bool lone_launch = false; // THIS launch of the lane belongs to a lone proof: its fold hops only to the lane's lone fold stream, if the plan made one
    Stream lone_fold_stream = 0;   // ("gl" / "zl" / "hl" of the stream plan: slot 0's lanes)
    bool lone_fold_made = false;
    bool share_cu = false;    // THIS launch of the lane: one accumulation workgroup per CU (LDS padding) at raised wave priority — a lone proof's
                              // G2 lane (zkhip_ctx::lone_sched); reset by msm_run_tables
};
static constexpr int ZK_NLANES = 5;   // A, B1, L (G1), B2 (G2) over z; H over h
static constexpr int ZK_NSLOTS = 4;   // most proofs in flight (zkhip_prove_g16*_batch pipelines consecutive proofs; ctx->nslots of them are used)
// everything one proof in flight owns: its scalars, NTT vectors, sort results, MSM workspaces, window sums and events
struct ProofSlot {
    DBuf scalars, zmont, va, vb, vc, ws1, ws2, zflag;   // zflag: one word, set by k_check_canonical when a host assignment is staged
    MsmSort sorts[3];         // over z (every table), over h, over z without the variables a family of tables holds at infinity (zkhip_pk::thin_mask)
    MsmLane lanes[ZK_NLANES];
    void* h_ws = nullptr;      // pinned host copy of the window sums
    size_t h_ws_cap = 0;
    Event ev[4] = {nullptr, nullptr, nullptr, nullptr};   // staged, MSMs over z issued, h ready, all done (copied out)
    Event acc_b[ZK_NLANES] = {}, acc_e[ZK_NLANES] = {};   // around each accumulation kernel
    Event ntt_b = nullptr, ntt_e = nullptr;               // around the transforms (7 for Groth16: 6 batched launches + 2; 5 for GM17)
    Event g1_go = nullptr;                                // a lone proof's G1 lanes held for the G2 accumulation (Prover::enqueue, g2_head_start)
    Event half_ready = nullptr;                           // a member of a multi-GPU proof: its half of the witness map (a or b on the coset) is in va
    int half = -1;                                        // which half this proof's head computed (-1: the whole witness map)
    bool lone = false;                                    // (enqueue_head -> enqueue_tail)
    Stream fold_slot = 0;                                 // one stream for the fold chains of this slot's three lanes ("f" of the stream plan)
    bool fold_slot_made = false;
    bool ready = false;        // streams and events exist (slot_init)
    // the proof currently in flight in this slot
    bool busy = false;
    uint8_t r[32], s[32];
    std::chrono::steady_clock::time_point t_start;
};
struct zkhip_ctx {
    int device = 0;
    Stream stream = 0;        // main stream: staging, sort, mat-vec, NTTs (high priority: short kernels the H MSM waits for)
    Stream out_stream = 0;    // copies the window sums out once every MSM of a proof is done
    Stream ntt_stream = 0;    // the mat-vec / NTT / h-sort pipeline of a proof in flight (high priority): off the main stream, so that
                              // the next proof's staging and z-sort (what its four big MSMs wait for) do not queue behind it
    bool out_made = false, ntt_made = false;   // (both made on first use: ctx_out_stream, ctx_ntt_stream)
    Stream ws = 0;            // the stream the mat-vec / NTT helpers launch on right now (ntt_stream inside a proof, else `stream`)
    bool serial = false;      // ZKHIP_SERIAL=1: every MSM on the main stream (debugging / per-kernel timing)
    bool g2_first = true;     // the G2 lane's stream at high priority (see slot_init)
    int cus = 256;            // compute units of the device: sizes the accumulation launch (one slice per resident work-item)
    // tunables (zkhip_ctx_tune; the environment variables ZKHIP_SERIAL, ZKHIP_MSM_C, ZKHIP_MSM_WAVES and
    // ZKHIP_NTT_SINGLE_MAX_LOG give their initial values, read ONCE when the context is created)
    int msm_c_env = 0;        // window width of the tables built / ad-hoc MSMs run from now on (0 = automatic)
    int b_sort_mode = 0;      // the thinned list (zkhip_pk::thin_mask): 0 when a tenth of a family's bases are at infinity, 1 always, 2 never
    int skip_inf_mode = 0;    // which accumulation kernel meets bases at infinity how: 0 per table (MsmShape::skip_inf), 1 lanes always sit them out, 2 always the vote
    int msm_sets = 0;         // bucket sets of the tables built from now on: 1 = every window multiple, 2 = every second ... (0 = what fits the device)
    int msm_waves = 0;        // accumulation waves per SIMD (0 = per point type)
    u32 msm_lanes = 0;        // slices of the sorted list (0 = one per resident work-item)
    u32 msm_min_slice = 8;    // finest cut of the sorted list
    bool heavy_runs = true;   // k_msm_heavy_reduce before the fold (ZKHIP_MSM_HEAVY_RUNS=0: the row's workgroup sums a heavy bucket alone)
    int heavy_threads = 64;   // work-items per workgroup of k_msm_heavy_reduce (64 / 128 / 256; ZKHIP_HEAVY_THREADS).  The kernel is launched whether or not
                              // a heavy bucket exists (the list is on the device) and must find a place before it can return: one wave and 9 / 18 KB of
                              // LDS (G1 / G2) instead of four and 36 / 73 is a place found sooner beside the accumulations, and a run of a heavy bucket is
                              // 8 serial additions + 6 tree levels instead of 2 + 8.  Lone dense proofs: median 9.61 against 9.78 ms over three processes
                              // of 40 each, the stdlib SHA-256 circuit (which HAS heavy buckets) and the others level (profiles/r7s_*, r7q_*)
    int fold_hg = 32;         // shares a column of rows is cut into in k_msm_fold_cols (a power of two <= 256; ZKHIP_FOLD_HG): 128 rows = 4 serial additions +
                              // 5 tree levels instead of 16 + 3 at 8 (the fold is a chain of dependent additions: Poseidon BLS12-381 8.95 -> 7.85 ms single)
    bool fold_scan = true;    // scan form of the last fold step (else double-and-add)
    int fold_lines = 0;       // rows and columns of the bucket matrix in one launch (k_msm_fold_lines): 0 never (the rows-then-columns pair), 1 always,
                              // 2 for launches over ONE table.  Measured and left OFF (profiles/r6m_fold_layouts_ab.txt): twice the workgroups of the
                              // row pass — over three tables (A, B1, L) they no longer fit the machine in one round (290 us against 178 + 108), over
                              // one table the launch saved is all there is (210 against 119 + 98); 110-112 proofs/s against 115-116 (ZKHIP_FOLD_LINES)
    bool fuse_z = true;       // A, B1 and L of a proof (one sorted list) as ONE slicing / accumulation / fold launch each
    int msm_fused_waves = 0;  // accumulation waves per SIMD of that launch (0 = per point type)
    int msm_g1_waves = 0, msm_g2_waves = 0;   // slices per SIMD lane of a single-table G1 / G2 accumulation (0 = per point type)
    int z_gate = 1;           // which accumulations over z wait for the witness map of their proof: 0 none, 1 the G1 lanes, 2 all
    int lone_sched = 0;       // how a LONE proof (the single-proof entry points; nothing else of the context in flight) is laid out — bits:
                              // 1: its G2 accumulation takes ONE workgroup per CU (dynamic LDS padded to more than half a CU's) at raised wave
                              //    priority: the short kernels of the witness map and of the sorts find a place at once instead of waiting for a
                              //    round of accumulation workgroups to retire (a 0.65 ms witness map took 3.4 ms beside a machine-filling G2
                              //    accumulation, the h sort 3.3 ms instead of 0.5: profiles/r6a_lone_bound_proof_gantt.txt);
                              // 2: its G1 lanes over z wait for the h SORT as well (else for the witness map only): that sort then runs beside
                              //    the G2 lane alone, and the H accumulation can start with the others
                              // (ZKHIP_LONE_SCHED / ZKHIP_TUNE_LONE_SCHED; batches are untouched)
    int g2_head_start = 1;    // a LONE proof over a curve whose G2 accumulation runs one wave per SIMD: its G1 lanes also wait for the end of
                              // that accumulation — 0 never, 1 over a bound key, 2 always (ZKHIP_G2_HEAD_START; Prover::enqueue)
    int split_min_log = 18;   // smallest domain (log2) whose witness map the members of a multi-GPU proof split between them (below: every
                              // member runs all of it — two transforms of 2^17 elements cost less than the exchange; ZKHIP_SPLIT_MIN_LOG)
    int ntt_single_max = 10;  // largest domain handled by one LDS-resident pass
    int ntt_max_sublog = 11;  // largest sub-transform of a pass (2^11 elements staged per sequence); domains above twice this take three passes
    int ntt_cols = 2;         // adjacent columns per workgroup of the cols pass (64-byte rows in HBM at 2)
    int fold_hop = 0;         // the fold chain of a lane on a stream (hardware queue) of its own behind the accumulation: 0 never, 1 every lane,
                              // 2 the G2 lane only (ZKHIP_FOLD_HOP).  A dispatch with workgroups still to place holds its queue's pipe; a fold
                              // that shares the pipe of a later accumulation waits for that accumulation's last round.
    int stream_skew = 0;      // streams made and left idle before a slot's lanes make theirs (ZKHIP_STREAM_SKEW): shifts which hardware queue a lane gets
    std::vector<Stream> skew_streams;
    std::string pipe_plan;    // which dispatcher ("pipe") each stream of a resident prover sits on (ZKHIP_PIPES; make_pipe_streams below)
    bool pipes_made = false;
    Stream ntt_lone_stream = 0;              // the witness map of a LONE proof ("n" of the plan), where the plan gives it a pipe of its own choosing
    bool ntt_lone_made = false;
    Stream fold_shared[3] = {0, 0, 0};       // the fold streams of the lanes Z, G, H when the plan gives one per type
    bool fold_shared_made[3] = {false, false, false};
    int ntt_fuse_first = 1;   // the first butterfly round of a pass done on the elements as they are fetched (kernels_ntt.cuh ntt_first_round; ZKHIP_NTT_FUSE_FIRST=0: every round through LDS)
    int ntt_skew_us = 0;      // start skew of a transform pass's first round of workgroups (kernels_ntt.cuh NttSkew; ZKHIP_NTT_SKEW_US)
    u32 sort_wgs = 256;       // workgroups of a counting-sort pass (chunks x windows): fewer = longer runs per (workgroup, bucket)
    int sort_two_level = 1;   // the placement pass of the sort in two levels (coarse bins of 256 buckets, then tiles: kernels_msm.cuh 1b); 0: round 5's
                              // one-level k_msm_place (ZKHIP_SORT_TWO_LEVEL)
    int sort_kh_log = 15;     // log2 of the counters of one sort workgroup's LDS histogram (ZKHIP_SORT_KH_LOG: development knob — a smaller
                              // histogram leaves LDS to the kernels beside it and reads the digits once more per halving)
    int tenants = 1;          // contexts of one zkhip_multi that share this context's device (their keys are sized for a share of its memory)
    int nslots = 3;           // proofs in flight in the batch calls (<= ZK_NSLOTS; measured 2 / 3 / 4: 72.0 / 76.0 / 74.8 proofs/s)
    std::string err;
    std::string desc;
    ProofSlot slots[ZK_NSLOTS];
    ProofSlot* cur = &slots[0];   // the slot the primitives (ntt, msm, ...) and the next enqueue work in
    DBuf tmp;
    std::vector<std::unique_ptr<NttPlanBase>> plans;
    // kernels whose dynamic-LDS limit has been raised for this context's device (the attribute is per device: a flag per
    // process would leave the second GPU of a process at the 64 KiB default)
    std::unordered_set<const void*> lds_opted;
};
// ---- streams by dispatcher.  The hardware queues of a process are dealt round-robin to the chip's four compute dispatchers
// ("pipes") in the order they are made, whatever their priority, and a dispatch that still has workgroups to place holds its
// pipe: a kernel that arrives on another queue of that pipe waits until the whole of it is placed — 0.3 ms behind a 0.6 ms
// kernel of four rounds, 3 ms behind an accumulation — while a kernel on another pipe gets the first place that comes free
// (tools/pipe_probe.hip, profiles/r6s_pipe_probe.txt).  A proof's fold chains, transforms and sorts are short kernels that
// arrive while accumulations are being placed; which pipe their streams share with which accumulation used to be an accident
// of the order of first use.  The plan names the pipe (0..3, relative to one another) of every stream of a resident prover:
//   M main (staging, z sort)  N witness map + h sort  O copy-out  |  per slot: Z, G, H the lanes of A/B1/L, B2 and H;
//   z, g, h a second stream of that lane TYPE for its fold chain (absent: the fold stays behind its accumulation);
//   f one stream per slot for the fold chains of its three lanes; gl / zl / hl a fold stream of slot 0's lane used by LONE proofs only
// e.g. "M=3,N=3,O=3,G=0,Z=1,H=2,g=3,z=3,h=3"; a name followed by a slot number ("G1=2") overrides that slot.  All streams of
// the plan are made in one go, in pipe order (idle streams fill the gaps), the first time a proof is enqueued.
// the plan of a resident prover (ZKHIP_TUNE_PIPE_PLAN = 1): found by a local search over the pipe of every stream, scored on four workloads at
// once — dense 2^20 BN254, stdlib SHA-256 2^20, the Poseidon chain on BLS12-381 2^18, GM17 2^20 — batches AND lone proofs (tools/plan_search.py,
// profiles/r7f_*, r7h_*, validated in r7g and at the end of r7h).  What its good layouts share: the witness-map stream N (and a lone proof's, n) on a pipe that
// holds no A/B1/L lane and no H lane, only G2 lanes of later slots; slot 0's three lanes — the ones a lone proof uses — never three on one pipe.
// Against streams in order of first use: dense level in batches and a lone proof 0.25-0.3 ms sooner, stdlib SHA-256 +8-9 % proofs/s, Poseidon
// +3 %, GM17 level; no workload slower.  (Round 6's first plan — one pipe per lane TYPE, fold streams per type: "M=1,N=2,n=3,O=0,G=0,Z=1,H=2,
// g=3,z=3,h=3" — gave the dense circuit the same and cost the thin ones 5-15 %: profiles/r6t_*, r6w_*.)  Still a request, not a default: the
// sixteen streams cost a one-proof process 0.15 s.
#define ZK_PIPE_PLAN_RESIDENT "M=2,N=3,O=1,n=3,G0=0,Z0=1,H0=0,G1=3,Z1=1,H1=0,G2=3,Z2=2,H2=1"
struct PipeWant { Stream* target; bool* made; bool high; int cls; bool done; };
static inline void make_pipe_streams(zkhip_ctx* ctx) {
    if (ctx->pipes_made || ctx->pipe_plan.empty() || ctx->serial) return;
    ctx->pipes_made = true;
    auto find = [&](const std::string& key) -> int {       // "key=d" in the plan, -1 if absent
        size_t pos = 0;
        const std::string& pl = ctx->pipe_plan;
        while (pos < pl.size()) {
            size_t end = pl.find(',', pos);
            if (end == std::string::npos) end = pl.size();
            const std::string tok = pl.substr(pos, end - pos);
            const size_t eq = tok.find('=');
            if (eq != std::string::npos && tok.substr(0, eq) == key && eq + 1 < tok.size() && tok[eq + 1] >= '0' && tok[eq + 1] <= '3') return tok[eq + 1] - '0';
            pos = end + 1;
        }
        return -1;
    };
    std::vector<PipeWant> want;
    static bool dummy_made;
    int c;
    if ((c = find("M")) >= 0) { stream_sync(ctx->stream); stream_destroy(ctx->stream); ctx->stream = 0; want.push_back({&ctx->stream, &dummy_made, true, c, false}); }
    if ((c = find("N")) >= 0 && !ctx->ntt_made) want.push_back({&ctx->ntt_stream, &ctx->ntt_made, true, c, false});
    if ((c = find("n")) >= 0) want.push_back({&ctx->ntt_lone_stream, &ctx->ntt_lone_made, true, c, false});
    if ((c = find("O")) >= 0 && !ctx->out_made) want.push_back({&ctx->out_stream, &ctx->out_made, false, c, false});
    static const char names[3] = {'Z', 'G', 'H'};
    static const int lane_of[3] = {0, 3, 4};
    for (int t = 0; t < 3; ++t) {       // a fold stream per lane type ("g=3"), unless the plan names slots ("g0=3")
        const std::string low(1, (char)(names[t] + 32));
        if ((c = find(low)) >= 0 && find(low + "0") < 0) want.push_back({&ctx->fold_shared[t], &ctx->fold_shared_made[t], true, c, false});
    }
    for (int k = 0; k < std::min(ctx->nslots, ZK_NSLOTS); ++k)
        for (int t = 0; t < 3; ++t) {
            MsmLane& lane = ctx->slots[k].lanes[lane_of[t]];
            const std::string up(1, names[t]), low(1, (char)(names[t] + 32)), num = std::to_string(k);
            c = find(up + num); if (c < 0) c = find(up);
            if (c >= 0 && !lane.made) want.push_back({&lane.stream, &lane.made, t == 1 && ctx->g2_first, c, false});
            c = find(low + num);
            if (c >= 0 && !lane.fold_made) want.push_back({&lane.fold_stream, &lane.fold_made, true, c, false});
        }
    for (int t = 0; t < 3; ++t) {       // slot 0's lanes in a LONE proof: "gl=3" — the fold of that lane hops to a stream of its own
        MsmLane& lane = ctx->slots[0].lanes[lane_of[t]];
        if ((c = find(std::string(1, (char)(names[t] + 32)) + "l")) >= 0 && !lane.lone_fold_made) want.push_back({&lane.lone_fold_stream, &lane.lone_fold_made, true, c, false});
    }
    for (int k = 0; k < std::min(ctx->nslots, ZK_NSLOTS); ++k) {
        c = find("f" + std::to_string(k)); if (c < 0) c = find("f");
        if (c >= 0 && !ctx->slots[k].fold_slot_made) want.push_back({&ctx->slots[k].fold_slot, &ctx->slots[k].fold_slot_made, true, c, false});
    }
    size_t left = want.size();
    for (int t = 0; left; ++t) {
        PipeWant* pick = nullptr;
        for (auto& w : want) if (!w.done && w.cls == (t & 3)) { pick = &w; break; }
        if (pick) { *pick->target = pick->high ? stream_create_high_priority() : stream_create(); *pick->made = true; pick->done = true; --left; }
        else ctx->skew_streams.push_back(stream_create_low_priority());      // (an idle queue of a priority nothing else uses)
    }
    for (int k = 0; k < ZK_NSLOTS; ++k)
        if (ctx->slots[k].fold_slot_made)
            for (int t = 0; t < 3; ++t) {
                MsmLane& lane = ctx->slots[k].lanes[lane_of[t]];
                if (!lane.fold_made) { lane.fold_stream = ctx->slots[k].fold_slot; lane.fold_made = true; lane.fold_owned = false; }
            }
    for (int t = 0; t < 3; ++t)
        if (ctx->fold_shared_made[t])
            for (int k = 0; k < ZK_NSLOTS; ++k) {
                MsmLane& lane = ctx->slots[k].lanes[lane_of[t]];
                if (!lane.fold_made) { lane.fold_stream = ctx->fold_shared[t]; lane.fold_made = true; lane.fold_owned = false; }
            }
    ctx->ws = ctx->stream;
}
// streams and events of one proof slot, made the first time the slot is used
static inline void slot_init(zkhip_ctx* ctx, ProofSlot& sl) {
    if (sl.ready) return;
    while ((int)ctx->skew_streams.size() < ctx->stream_skew) ctx->skew_streams.push_back(stream_create());
    for (auto& so : sl.sorts) so.ready = event_create();
    for (int k = 0; k < ZK_NLANES; ++k) {
        // lane 3 is the G2 MSM: the longest accumulation AND the longest fold tail of a proof; at high priority its
        // workgroups are dispatched first, it finishes early and its tail hides under the G1 accumulations
        sl.lanes[k].high_priority = k == 3 && ctx->g2_first;       // (the stream itself is made when a launch first needs it: lane_stream)
        sl.lanes[k].done = event_create();
        sl.lanes[k].acc_done = event_create();
        sl.acc_b[k] = event_create();
        sl.acc_e[k] = event_create();
    }
    for (auto& e : sl.ev) e = event_create();
    sl.ntt_b = event_create();
    sl.ntt_e = event_create();
    sl.g1_go = event_create();
    sl.half_ready = event_create();
    sl.ready = true;
}
// streams made when they are first needed
static inline Stream lane_stream(MsmLane& lane) {
    if (!lane.made) { lane.stream = lane.high_priority ? stream_create_high_priority() : stream_create(); lane.made = true; }
    return lane.stream;
}
static inline Stream lane_fold_stream(MsmLane& lane) {
    if (!lane.fold_made) { lane.fold_stream = stream_create_high_priority(); lane.fold_made = true; }
    return lane.fold_stream;
}
static inline Stream ctx_out_stream(zkhip_ctx* ctx) {
    if (!ctx->out_made) { ctx->out_stream = stream_create(); ctx->out_made = true; }
    return ctx->out_stream;
}
static inline Stream ctx_ntt_stream(zkhip_ctx* ctx);
// the stream of a proof's witness map: the context's, or — a lone proof under a plan that names one — the lone proofs' own
static inline Stream slot_ntt_stream(zkhip_ctx* ctx, const ProofSlot& sl) {
    return (sl.lone && ctx->ntt_lone_made) ? ctx->ntt_lone_stream : ctx_ntt_stream(ctx);
}
static inline Stream ctx_ntt_stream(zkhip_ctx* ctx) {
    if (!ctx->ntt_made) { ctx->ntt_stream = stream_create_high_priority(); ctx->ntt_made = true; }
    return ctx->ntt_stream;
}
// gfx950 has 160 KiB of LDS per CU; anything above the 64 KiB default must be opted into, per kernel and per device
// the z-lane gate of the proof being enqueued (see Prover::enqueue)
static inline int z_gate(const zkhip_ctx* ctx) { return ctx->serial ? 0 : ctx->z_gate; }
static inline void lds_opt_in(zkhip_ctx* ctx, const void* kernel) {
#ifndef ZK_EMU
    if (ctx->lds_opted.count(kernel)) return;
    ZK_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->lds_opted.insert(kernel);
#else
    (void)ctx; (void)kernel;
#endif
}

namespace zk {

// ------------------------------------------------------------------ NTT plan
template <class C>
struct NttPlan : NttPlanBase {
    typedef typename C::Fr Fr;
    typedef Fu<typename Fr::Params> FrU;   // the passes' working form (kernels_ntt.cuh)
    int log1, log2, log3;    // N = N1 * N2 * N3: cols pass over N1, (three passes, log3 > 0: cols pass over N2 inside every N2 x N3 block,) rows pass over the last factor
    u64 N;
    u32 N1, N2, N3, M;       // M = max(N1, N2, N3): root tables hold w_M^j, j < M; N3 = 1 for one or two passes
    u64 Nb;                  // N2 * N3: the block of one outer index
    int split() const { return log1 | (log3 << 8); }   // what the sigma order depends on besides N (zkhip_pk::ntt_log1, key images)
    // roots: 9-limb R'-form; tw (inter-pass twiddles, the inverse ones times 1/N... see get_plan), s_coset (g^i / N, sigma
    // order), s_cosetinv_canon (g^-i / N as plain integers: the Montgomery exit): packed, N entries each
    DBuf roots_fwd, roots_inv, tw_fwd, tw_inv, s_coset, s_cosetinv_canon;
    DBuf s_cexit;            // zinv / N as plain integers, N entries: exit factor of the transform that turns c's evaluations into
                             // its share zinv * c_i of the quotient's coefficients (canonical integers)
    DBuf tw2_fwd, tw2_inv;   // three passes: the twiddles inside a block, w_Nb^(k2 * j3), Nb entries
    DBuf plan1[2], plan2[2], plan3[2];   // twiddle plans of the N1-, N2- and N3-point sub-NTTs, [0] forward, [1] inverse (kernels_ntt.cuh)
    u32 plen1 = 0, plen2 = 0, plen3 = 0;
    Fr omega, omega_inv, n_inv, g, g_inv, zinv;   // saturated Montgomery form (host code, setup)
    Fr zinv_rp;                                   // zinv in R'-form (k_quotient)
    Fr k_to_rp, k_mont_to_rp;                     // canonical integer -> R'-form, saturated Montgomery -> R'-form (fe_mul by these)
    int C_cols, R_rows, threads_cols, threads_rows;
    size_t smem_cols, smem_rows;
    int C_mid = 1, threads_mid = 64;   // the middle pass of three: columns per workgroup, work-items
    size_t smem_mid = 0;
};

template <class Fr>
static Fr host_root_of_unity(int two_adicity, u64 generator, int logN) {
    // GENERATOR^((r-1) >> S), then squared down to order 2^logN   (App. A.4)
    u32 e[Fr::N];
    u64 bw = 1;
    for (int i = 0; i < Fr::N; ++i) {
        u64 t = (u64)Fr::Params::mod(i) - bw;
        e[i] = (u32)t;
        bw = t >> 63;
    }
    for (int s = 0; s < two_adicity; ++s)
        for (int i = 0; i < Fr::N; ++i) e[i] = (e[i] >> 1) | (i + 1 < Fr::N ? e[i + 1] << 31 : 0);
    Fr root = fe_pow(fe_from_u64<typename Fr::Params>(generator), e, Fr::N);
    for (int i = logN; i < two_adicity; ++i) root = fe_sqr(root);
    return root;
}

static constexpr int NTT_MAX_SUBLOG = 11;   // 2^11 x 32 B = 64 KiB per staged sequence

template <class C>
static NttPlan<C>* get_plan(zkhip_ctx* ctx, int logN) {
    typedef typename C::Fr Fr;
    for (auto& p : ctx->plans)
        if (p->curve == C::ID && p->logN == logN) return (NttPlan<C>*)p.get();
    const int sub = ctx->ntt_max_sublog;
    require(logN >= 0 && logN <= 3 * sub && logN <= C::TWO_ADICITY, ZKHIP_ERR_BAD_ARG,
            "domain size unsupported (log2 N must not exceed the field's two-adicity: 28 for bn128, 32 for bls12_381)");
    auto* pl = new NttPlan<C>();
    ctx->plans.emplace_back(pl);
    pl->curve = C::ID;
    pl->logN = logN;
    pl->N = (u64)1 << logN;
    // one pass up to 2^ntt_single_max, two passes (N1 x N2) up to 2^(2 sub), three (N1 x N2 x N3) beyond: the reference's
    // radix-2 domain goes up to the field's two-adicity (ark-poly Radix2EvaluationDomain), so does this one
    if (logN <= ctx->ntt_single_max && logN <= sub) { pl->log1 = 0; pl->log2 = logN; pl->log3 = 0; }
    else if (logN <= 2 * sub) { pl->log1 = logN / 2; pl->log2 = logN - pl->log1; pl->log3 = 0; }
    else { pl->log1 = logN / 3; pl->log2 = (logN - pl->log1) / 2; pl->log3 = logN - pl->log1 - pl->log2; }
    pl->N1 = 1u << pl->log1;
    pl->N2 = 1u << pl->log2;
    pl->N3 = 1u << pl->log3;
    pl->Nb = (u64)pl->N2 * pl->N3;
    pl->M = std::max(pl->N1, std::max(pl->N2, pl->N3));
    pl->omega = host_root_of_unity<Fr>(C::TWO_ADICITY, C::GENERATOR, logN);
    pl->omega_inv = fe_inv(pl->omega);
    pl->n_inv = fe_inv(fe_from_u64<typename Fr::Params>(pl->N));
    pl->g = fe_from_u64<typename Fr::Params>(C::GENERATOR);
    pl->g_inv = fe_inv(pl->g);
    pl->zinv = fe_inv(fe_sub(fe_pow_u64(pl->g, pl->N), Fr::one()));
    Stream s = ctx->stream;
    const unsigned T = 256;
    // R' = 2^(29*9) as a canonical integer mod p: Fr::one() is the integer 2^256 mod p, doubled 5 more times
    Fr rp = Fr::one();
    for (int i = 32 * Fr::N; i < Fu<typename Fr::Params>::B * Fu<typename Fr::Params>::N; ++i) rp = fe_add(rp, rp);
    pl->k_mont_to_rp = rp;                    // fe_mul(x R, R') = x R'
    pl->k_to_rp = fe_to_mont(rp);             // fe_mul(x, R' R) = x R'
    pl->zinv_rp = fe_mul(pl->zinv, rp);
    typedef typename NttPlan<C>::FrU FrU;
    auto to_rp = [&](DBuf& buf, u64 count) {  // saturated Montgomery table -> packed R'-form, in place
        ZK_LAUNCH((k_mul_const<Fr>), dim3(blocks_for(count, T)), dim3(T), 0, s, ptr<Fr>(buf), ptr<Fr>(buf), count, rp);
    };
    // sub-NTT roots: w_M^j, j < M (the radix-4 butterflies use w^pos, w^2pos, w^3pos), as 9-limb R'-form values
    const Fr wM = fe_pow_u64(pl->omega, pl->N / pl->M);
    const u64 nroots = std::max<u64>(pl->M, 1);
    ctx->tmp.ensure(nroots * sizeof(Fr));
    for (int inv = 0; inv < 2; ++inv) {
        DBuf& dst = inv ? pl->roots_inv : pl->roots_fwd;
        dst.ensure(nroots * sizeof(FrU));
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(nroots, T)), dim3(T), 0, s, ptr<Fr>(ctx->tmp), inv ? fe_inv(wM) : wM, Fr::one(), nroots, 0u, 0u, 0, 1u);
        to_rp(ctx->tmp, nroots);
        ZK_LAUNCH((k_unpack_table<typename Fr::Params>), dim3(blocks_for(nroots, T)), dim3(T), 0, s, ptr<Fr>(ctx->tmp), ptr<FrU>(dst), nroots);
    }
    // the sub-NTTs' twiddle plans, gathered from the root tables
    {
        std::vector<u32> src;
        DBuf d_src;
        for (int which = 0; which < 3; ++which) {
            if (which == 2 && pl->log3 == 0) break;
            const int lg = which == 0 ? pl->log1 : which == 1 ? pl->log2 : pl->log3;
            const u32 plen = ntt_plan_len(lg);
            (which == 0 ? pl->plen1 : which == 1 ? pl->plen2 : pl->plen3) = plen;
            ntt_plan_exponents(lg, src);
            d_src.ensure(src.size() * 4);
            dev_h2d(d_src.p, src.data(), src.size() * 4, s);
            for (int inv = 0; inv < 2; ++inv) {
                DBuf& dst = which == 0 ? pl->plan1[inv] : which == 1 ? pl->plan2[inv] : pl->plan3[inv];
                dst.ensure((size_t)plen * NTT_PLAN_STRIDE * 4);
                ZK_LAUNCH((k_ntt_plan_gather<typename Fr::Params>), dim3(blocks_for(plen, T)), dim3(T), 0, s, ptr<FrU>(inv ? pl->roots_inv : pl->roots_fwd),
                          (int)(pl->M >> lg), ptr<u32>(d_src), plen, ptr<u32>(dst));
            }
            stream_sync(s);   // src is reused
        }
    }
    if (pl->log1 > 0) {
        // between the pass over N1 and what follows: w_N^(k1 * j), j < Nb the position inside the block of k1
        pl->tw_fwd.ensure(pl->N * sizeof(Fr));
        pl->tw_inv.ensure(pl->N * sizeof(Fr));
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->N, T)), dim3(T), 0, s, ptr<Fr>(pl->tw_fwd), pl->omega, Fr::one(), pl->N, pl->N1, (u32)pl->Nb, 1, 1u);
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->N, T)), dim3(T), 0, s, ptr<Fr>(pl->tw_inv), pl->omega_inv, Fr::one(), pl->N, pl->N1, (u32)pl->Nb, 1, 1u);
        to_rp(pl->tw_fwd, pl->N);
        to_rp(pl->tw_inv, pl->N);
    }
    if (pl->log3 > 0) {
        // inside a block: w_Nb^(k2 * j3), the two-pass rule once more (w_Nb = w_N^N1)
        const Fr wb = fe_pow_u64(pl->omega, pl->N1), wb_inv = fe_pow_u64(pl->omega_inv, pl->N1);
        pl->tw2_fwd.ensure(pl->Nb * sizeof(Fr));
        pl->tw2_inv.ensure(pl->Nb * sizeof(Fr));
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->Nb, T)), dim3(T), 0, s, ptr<Fr>(pl->tw2_fwd), wb, Fr::one(), pl->Nb, pl->N2, pl->N3, 1, 1u);
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->Nb, T)), dim3(T), 0, s, ptr<Fr>(pl->tw2_inv), wb_inv, Fr::one(), pl->Nb, pl->N2, pl->N3, 1, 1u);
        to_rp(pl->tw2_fwd, pl->Nb);
        to_rp(pl->tw2_inv, pl->Nb);
    }
    pl->s_coset.ensure(pl->N * sizeof(Fr));
    pl->s_cosetinv_canon.ensure(pl->N * sizeof(Fr));
    ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->N, T)), dim3(T), 0, s, ptr<Fr>(pl->s_coset), pl->g, pl->n_inv, pl->N, pl->N1, pl->N2, 2, pl->N3);
    to_rp(pl->s_coset, pl->N);
    // plain integers: multiplying an R'-form value by them (R' Montgomery product) leaves the plain value
    ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->N, T)), dim3(T), 0, s, ptr<Fr>(pl->s_cosetinv_canon), pl->g_inv, fe_from_mont(pl->n_inv), pl->N,
              pl->N1, pl->N2, 2, pl->N3);
    pl->s_cexit.ensure(pl->N * sizeof(Fr));
    ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(pl->N, T)), dim3(T), 0, s, ptr<Fr>(pl->s_cexit), Fr::one(), fe_from_mont(fe_mul(pl->n_inv, pl->zinv)), pl->N,
              0u, 0u, 0, 1u);
    // launch geometry.  A workgroup stages `tile` elements as nine limb planes (36 B + padding per element): tiles of
    // 1024 elements (38 KiB) let four workgroups share a CU, so that one loads while another computes; a cols tile
    // needs >= 2 columns for 64-byte rows in HBM.
    const u32 last = pl->log3 > 0 ? pl->N3 : pl->N2;                        // length of the rows pass's sequences
    const u64 nrows = pl->N / last;
    const u32 tile_rows = std::max<u32>(last, (u32)std::min<u64>(pl->N, 1024));
    u32 R = pl->log1 == 0 ? 1 : (u32)std::max<u64>(1, std::min<u64>(nrows, tile_rows / last));
    pl->R_rows = (int)R;
    const u32 ccols = (u32)ctx->ntt_cols;
    auto cols_per_wg = [&](u64 ncols, u32 n) { return (int)std::max<u64>(1, std::min<u64>(ncols, std::max<u32>(ccols, 1024 / std::max<u32>(n, 1)))); };
    pl->C_cols = cols_per_wg(pl->Nb, pl->N1);
    require(pl->Nb % (u64)pl->C_cols == 0 && nrows % R == 0, ZKHIP_ERR_BAD_ARG, "internal: NTT tile does not divide the domain");
    auto smem_for = [](u32 nseq, u32 n) { return (size_t)9 * nseq * ntt_seq_stride((int)n) * 4; };
    pl->smem_cols = smem_for(pl->C_cols, pl->N1);
    pl->smem_rows = smem_for(R, last);
    auto pick_threads = [](u64 butterflies) { return (int)std::min<u64>(512, std::max<u64>(64, (butterflies + 63) / 64 * 64)); };
    pl->threads_cols = pick_threads((u64)pl->C_cols * pl->N1 / 4);
    pl->threads_rows = pick_threads((u64)R * last / 4);
    if (pl->log3 > 0) {
        pl->C_mid = cols_per_wg(pl->N3, pl->N2);
        require(pl->N3 % (u32)pl->C_mid == 0, ZKHIP_ERR_BAD_ARG, "internal: NTT tile does not divide the domain");
        pl->smem_mid = smem_for(pl->C_mid, pl->N2);
        pl->threads_mid = pick_threads((u64)pl->C_mid * pl->N2 / 4);
        require(pl->N1 <= 65535, ZKHIP_ERR_BAD_ARG, "internal: NTT batch exceeds the grid");
    }
    lds_opt_in(ctx, (const void*)k_ntt_cols<typename Fr::Params>);
    lds_opt_in(ctx, (const void*)k_ntt_rows<typename Fr::Params>);
    stream_sync(s);
    return pl;
}

// the start skew of a pass whose workgroups take `smem` bytes of LDS each: the first round = as many workgroups as the device holds
static inline NttSkew ntt_skew(const zkhip_ctx* ctx, size_t smem) {
    if (ctx->ntt_skew_us <= 0) return NttSkew{0, 0};
    const u32 per_cu = (u32)std::max<size_t>(1, std::min<size_t>(4, (size_t)160 * 1024 / std::max<size_t>(smem, 1)));
    return NttSkew{(u32)ctx->ntt_skew_us * 100u, (u32)ctx->cus * per_cu};
}
// `nvec` vectors of N elements, vec_stride elements apart, go through one launch (grid.y)
// the pass over N1: columns of the N1 x Nb matrix
template <class C>
static void ntt_cols(zkhip_ctx* ctx, NttPlan<C>* pl, typename C::Fr* data, bool inverse, const typename C::Fr* post, int nvec = 1, u64 vec_stride = 0,
                     int canon = 0, const typename C::Fr* minus = nullptr) {
    typedef typename C::Fr Fr;
    ZK_LAUNCH((k_ntt_cols<typename Fr::Params>), dim3((unsigned)(pl->Nb / pl->C_cols), nvec), dim3(pl->threads_cols), pl->smem_cols, ctx->ws, data, vec_stride,
              pl->log1, (u32)pl->Nb, pl->C_cols, ptr<u32>(pl->plan1[inverse ? 1 : 0]), pl->plen1, post, canon, minus, (u64)0, ~(u64)0, ntt_skew(ctx, pl->smem_cols), ctx->ntt_fuse_first);
}
// three passes only — the pass over N2: columns of the N2 x N3 matrix of every outer index (grid.z).  `post_mask`: Nb - 1 for
// the per-block twiddles, all ones for a table as long as the vector.
template <class C>
static void ntt_mid(zkhip_ctx* ctx, NttPlan<C>* pl, typename C::Fr* data, bool inverse, const typename C::Fr* post, u64 post_mask, int nvec, u64 vec_stride) {
    typedef typename C::Fr Fr;
    ZK_LAUNCH((k_ntt_cols<typename Fr::Params>), dim3(pl->N3 / pl->C_mid, nvec, pl->N1), dim3(pl->threads_mid), pl->smem_mid, ctx->ws, data, vec_stride,
              pl->log2, pl->N3, pl->C_mid, ptr<u32>(pl->plan2[inverse ? 1 : 0]), pl->plen2, post, 0, (const Fr*)nullptr, pl->Nb, post_mask, ntt_skew(ctx, pl->smem_mid), ctx->ntt_fuse_first);
}
// the pass over the last factor: contiguous sequences
template <class C>
static void ntt_rows(zkhip_ctx* ctx, NttPlan<C>* pl, typename C::Fr* data, bool inverse, const typename C::Fr* post, int nvec = 1, u64 vec_stride = 0,
                     int canon = 0, const typename C::Fr* minus = nullptr, u64 post_mask = ~(u64)0) {
    typedef typename C::Fr Fr;
    const bool three = pl->log3 > 0;
    const int lg = three ? pl->log3 : pl->log2;
    ZK_LAUNCH((k_ntt_rows<typename Fr::Params>), dim3((unsigned)((pl->N >> lg) / pl->R_rows), nvec), dim3(pl->threads_rows), pl->smem_rows, ctx->ws, data, vec_stride,
              lg, pl->R_rows, ptr<u32>((three ? pl->plan3 : pl->plan2)[inverse ? 1 : 0]), three ? pl->plen3 : pl->plen2, post, canon, minus, post_mask,
              ntt_skew(ctx, pl->smem_rows), ctx->ntt_fuse_first);
}
// natural order in -> sigma order out
template <class C>
static void ntt_kind_a(zkhip_ctx* ctx, NttPlan<C>* pl, typename C::Fr* data, bool inverse, const typename C::Fr* final_post, int nvec = 1,
                       u64 vec_stride = 0, int canon = 0, const typename C::Fr* final_minus = nullptr) {
    typedef typename C::Fr Fr;
    if (pl->log1 > 0) ntt_cols<C>(ctx, pl, data, inverse, inverse ? ptr<Fr>(pl->tw_inv) : ptr<Fr>(pl->tw_fwd), nvec, vec_stride);
    if (pl->log3 > 0) ntt_mid<C>(ctx, pl, data, inverse, inverse ? ptr<Fr>(pl->tw2_inv) : ptr<Fr>(pl->tw2_fwd), pl->Nb - 1, nvec, vec_stride);
    ntt_rows<C>(ctx, pl, data, inverse, final_post, nvec, vec_stride, canon, final_minus);
}
// sigma order in -> natural order out
template <class C>
static void ntt_kind_b(zkhip_ctx* ctx, NttPlan<C>* pl, typename C::Fr* data, bool inverse, const typename C::Fr* final_post, int nvec = 1,
                       u64 vec_stride = 0, int canon = 0) {
    typedef typename C::Fr Fr;
    if (pl->log3 > 0) {
        ntt_rows<C>(ctx, pl, data, inverse, inverse ? ptr<Fr>(pl->tw2_inv) : ptr<Fr>(pl->tw2_fwd), nvec, vec_stride, 0, nullptr, pl->Nb - 1);
        ntt_mid<C>(ctx, pl, data, inverse, inverse ? ptr<Fr>(pl->tw_inv) : ptr<Fr>(pl->tw_fwd), ~(u64)0, nvec, vec_stride);
        ntt_cols<C>(ctx, pl, data, inverse, final_post, nvec, vec_stride, canon);
    } else if (pl->log1 > 0) {
        ntt_rows<C>(ctx, pl, data, inverse, inverse ? ptr<Fr>(pl->tw_inv) : ptr<Fr>(pl->tw_fwd), nvec, vec_stride);
        ntt_cols<C>(ctx, pl, data, inverse, final_post, nvec, vec_stride, canon);
    } else {
        ntt_rows<C>(ctx, pl, data, inverse, final_post, nvec, vec_stride, canon);
    }
}

// ------------------------------------------------------------------ MSM driver
struct MsmShape {
    u64 n;
    int c, W;
    u32 K;          // buckets per set = 2^(c-1)
    u32 sets;       // bucket sets: 1 when the bases carry every window multiple (all windows share one set), W without a table, in
                    // between for keys too large for W levels (window j: bucket set j % sets, table level j / sets)
    u32 levels;     // table levels the sorted entries refer to: ceil(W / sets) (1 without a table)
    u32 nkeys;      // sets * K
    u32 Lw, H;      // fold geometry: K = H rows of Lw buckets
    static constexpr u32 ndig = 2;   // digits of the fold: column and row of the bucket index
    int level_bits() const { return c * (int)sets; }   // level t of a table holds 2^(level_bits t) P
    bool skip_inf = true;   // which accumulation kernel: lanes sit out bases at infinity (tables that hold many: kernels_msm.cuh
                            // k_msm_accum SKIP_INF), or the rare one goes through the general code (tables that hold next to none)
    u32 nsums() const { return ndig * sets; }   // the fold leaves one sum per digit and bucket set: msm_combine adds them
};
static inline int env_int(const char* name, int lo, int hi, int dflt) {
    if (const char* e = getenv(name)) { int v = atoi(e); if (v >= lo && v <= hi) return v; }
    return dflt;
}
static constexpr int MSM_MAX_C = 17;         // widest window: 2^16 buckets, which the sort takes in two halves of MSM_SORT_MAX_KH counters
static constexpr int MSM_AUTO_MAX_C = 17;    // widest window chosen automatically (resident tables)
static constexpr int MSM_ADHOC_MAX_C = 16;   // ... and without a table (a bucket set per window: the fold grows with every bucket)
static constexpr u32 MSM_SORT_MAX_KH = 1u << 15;   // counters of one sort workgroup's LDS histogram (128 KiB)
// `table`: the bases carry precomputed window multiples 2^(c j) P (resident keys); otherwise one bucket set per window.
static inline MsmShape msm_shape(const zkhip_ctx* ctx, u64 n, int scalar_bits, bool table, int force_c = 0, int force_sets = 0) {
    MsmShape s;
    s.n = n;
    const int lg = ilog2_floor(std::max<u64>(n, 1));
    if (table) {
        // one fold per MSM whatever c is, so c only trades additions (n * W) against buckets (2^(c-1)): the fewest windows the
        // widest admissible width gives (a few points per bucket at least), and of the widths with that many windows the
        // narrowest — 254-bit scalars: 15 windows of 17 bits instead of 16 of 16 (6 % fewer additions in all five MSMs for twice
        // the buckets); 255-bit scalars need 16 windows either way and stay at 16 bits
        const int cmax = std::max(2, std::min(MSM_AUTO_MAX_C, lg + 1));
        const int Wmin = (scalar_bits + 1 + cmax - 1) / cmax;
        s.c = cmax;
        while (s.c > 2 && (scalar_bits + 1 + (s.c - 1) - 1) / (s.c - 1) == Wmin) --s.c;
    } else {
        // Window width: about log2(n) - 5 (measured optimum at 2^20: 15), but never one that leaves the top window
        // with only a few significant bits — its handful of buckets would each receive a large share of all points
        // (same-address atomics in the sort, one bucket spread over thousands of slices).
        const int want = std::max(2, std::min(MSM_ADHOC_MAX_C, lg - 5));
        s.c = want;
        for (int d = 0; d <= 14; ++d) {
            bool found = false;
            for (int cand : {want + d, want - d}) {
                if (cand < 2 || cand > MSM_ADHOC_MAX_C) continue;
                const int W = (scalar_bits + 1 + cand - 1) / cand;
                const int top_bits = scalar_bits + 1 - (W - 1) * cand;
                if (top_bits >= cand || (n >> (top_bits - 1)) <= 4096) { s.c = cand; found = true; break; }   // <= 4096 points per top bucket
            }
            if (found) break;
        }
    }
    if (ctx && ctx->msm_c_env) s.c = ctx->msm_c_env;
    if (force_c) s.c = force_c;
    s.c = std::min(s.c, MSM_MAX_C);
    s.W = (scalar_bits + 1 + s.c - 1) / s.c;
    s.K = 1u << (s.c - 1);
    s.sets = table ? (u32)std::max(1, std::min(force_sets ? force_sets : 1, s.W)) : (u32)s.W;
    s.levels = ((u32)s.W + s.sets - 1) / s.sets;
    s.nkeys = s.sets * s.K;
    s.Lw = std::min<u32>(s.K, 256);
    s.H = s.K / s.Lw;
    return s;
}

static inline MsmShape with_inf(MsmShape sh, bool many) { sh.skip_inf = many; return sh; }
// Device memory the window-multiple tables of a key may take: 60 % of what is free now, less what the proofs themselves will
// allocate later beside them — per proof slot the digits and sorted lists of up to three sorts, the transform vectors, scalars, and
// the partial / bucket arrays of five MSMs — and the transient of the table construction (msm_table_levels: < 1 GiB).  Several
// contexts sharing one device each see the same "free" figure: the margin is what keeps the first batch prove from failing where
// the key load succeeded.
static inline u64 msm_table_budget(const zkhip_ctx* ctx, u64 z_n, u64 h_n, u64 N, int W, u32 K) {
    size_t free_b = 0, total_b = 0;
    dev_mem_info(&free_b, &total_b);
    const u64 per_slot = (2 * z_n + h_n) * (u64)W * 16 + 4 * N * 32 + (z_n + 2) * 64 + 6 * ((u64)K + (1u << 19)) * 288 + ((u64)1 << 29);
    const u64 later = (u64)std::max(1, ctx->nslots) * per_slot + ((u64)1 << 30);
    const u64 budget = (u64)((ctx->tenants > 1 ? 0.5 : 0.6) * (double)free_b) / (u64)std::max(1, ctx->tenants);
    return budget > later ? budget - later : 0;
}
// exclusive scan of nk counters (cnt -> off, off[nk] = their sum, also left in *grand)
static inline void scan_u32(Stream s, const DBuf& cnt, DBuf& off, u64 nk, DBuf& chunk_sum, DBuf& grand) {
    const u32 nchunks = (u32)((nk + SCAN_CHUNK - 1) / SCAN_CHUNK);
    off.ensure((nk + 1) * 4);
    chunk_sum.ensure((size_t)nchunks * 4);
    grand.ensure(4);
    ZK_LAUNCH(k_scan_local, dim3(nchunks), dim3(SCAN_THREADS), 0, s, ptr<u32>(cnt), ptr<u32>(off), ptr<u32>(chunk_sum), nk);
    ZK_LAUNCH(k_scan_chunks, dim3(1), dim3(SCAN_THREADS), 0, s, ptr<u32>(chunk_sum), nchunks, ptr<u32>(grand));
    ZK_LAUNCH(k_scan_add, dim3(blocks_for(nk + 1, 256)), dim3(256), 0, s, ptr<u32>(off), ptr<u32>(chunk_sum), nk, ptr<u32>(grand));
}
// digits + counting sort on the main stream; leaves so.off / so.sorted describing every bucket's point list.
// level_stride: distance between two levels of the base tables this sort will be paired with (table mode), else 0.
// `keep`: bitmap of the scalars that take part (null: all); `wm_of`: a sort of the SAME scalars (same shape) whose digits are reused
static inline void msm_prepare(zkhip_ctx* ctx, Stream s, MsmSort& so, const u32* d_scalars, const MsmShape& sh, u64 level_stride,
                               const u32* keep = nullptr, const MsmSort* wm_of = nullptr) {
    const u64 nk = sh.nkeys;
    require(sh.n * (u64)sh.W < ((u64)1 << 32) - sh.nkeys, ZKHIP_ERR_BAD_ARG, "MSM too large for 32-bit sort offsets");
    require(level_stride * (u64)sh.levels < ((u64)1 << 31) && sh.n < ((u64)1 << 31), ZKHIP_ERR_BAD_ARG, "MSM too large for 31-bit table indices");
    if (!wm_of) so.dig.ensure(sh.n * (u64)sh.W * 4);
    so.sorted.ensure(sh.n * sh.W * 4);
    const unsigned T = 256;
    // one workgroup per (chunk of scalars, window): chunks several times larger than a window's bucket count keep the
    // global atomics (one per touched bucket per workgroup) well below one per digit
    // (a window of more than 16 bits has more buckets than one histogram holds: its workgroups come in `halves`, each reading the
    // chunk's digits and keeping the ones of its own range of buckets)
    const u32 kh = std::min(sh.K, std::min(MSM_SORT_MAX_KH, 1u << ctx->sort_kh_log)), halves = sh.K / kh;
    const u64 want_chunks = halves > 1 ? std::max<u64>(1, ctx->sort_wgs / (sh.W * halves)) : std::max<u64>(1, (ctx->sort_wgs + sh.W - 1) / sh.W);
    const u64 max_chunks = std::max<u64>(1, sh.n / (2 * (u64)kh));
    const u64 sort_chunks = std::min(want_chunks, max_chunks);
    const u64 chunk = (sh.n + sort_chunks - 1) / sort_chunks;
    const size_t hist_bytes = (size_t)kh * 4;
    const bool two_level = ctx->sort_two_level && sh.K >= (1u << MSM_COARSE_BITS) && sh.K <= (1u << 16);
    const u32 nbins = (u32)(nk >> MSM_COARSE_BITS);
    so.cnt.ensure(nk * 4);
    so.cursor.ensure(nk * 4);
    if (two_level) {
        so.ccur.ensure((size_t)nbins * 4);
        so.tile_off.ensure(((size_t)nbins + 1) * 4);
    }
    // the counters of the passes below start from zero: cleared by the digit kernel on its way (a sort that borrows another's digits:
    // by a launch of its own)
    const MsmZero zero{{ptr<u32>(so.cnt), ptr<u32>(so.cursor), two_level ? ptr<u32>(so.ccur) : nullptr}, {(u32)nk, (u32)nk, two_level ? nbins : 0u}};
    if (!wm_of) ZK_LAUNCH(k_msm_digits, dim3(blocks_for(sh.n, T)), dim3(T), 0, s, d_scalars, sh.n, sh.c, sh.W, ptr<u32>(so.dig), zero);
    else ZK_LAUNCH(k_msm_zero, dim3(blocks_for(nk, T)), dim3(T), 0, s, zero);
    const u32* wm = wm_of ? ptr<u32>(wm_of->dig) : ptr<u32>(so.dig);
    lds_opt_in(ctx, (const void*)k_msm_count);
    lds_opt_in(ctx, (const void*)k_msm_place);
    ZK_LAUNCH(k_msm_count, dim3((unsigned)sort_chunks, sh.W, halves), dim3(ZK_SORT_THREADS), hist_bytes, s, wm, sh.n, sh.c, sh.W, chunk, sh.sets, kh,
              ptr<u32>(so.cnt), keep);
    if (nk <= SCAN_ONE_MAX) {
        so.off.ensure((nk + 1) * 4);
        ZK_LAUNCH(k_scan_one, dim3(1), dim3(SCAN_ONE_THREADS), 0, s, ptr<u32>(so.cnt), ptr<u32>(so.off), (u32)nk, two_level ? ptr<u32>(so.tile_off) : (u32*)nullptr, nbins);
    } else {
        scan_u32(s, so.cnt, so.off, nk, so.chunk_sum, so.grand);
        if (two_level) ZK_LAUNCH(k_msm_tile_offsets, dim3(1), dim3(MSM_TILE_SCAN_THREADS), 0, s, ptr<u32>(so.off), nbins, ptr<u32>(so.tile_off));
    }
    if (two_level) {
        const u64 total_max = sh.n * (u64)sh.W;
        so.pairs.ensure(std::max<u64>(total_max, 1) * 8);
        // ~1024 workgroups of the coarse pass (1 KiB of LDS each: they share CUs with anything), chunks of at least 4096 scalars
        const u64 c_chunks = std::max<u64>(1, std::min<u64>((1024 + sh.W - 1) / sh.W, (sh.n + 4095) / 4096));
        const u64 c_chunk = (sh.n + c_chunks - 1) / c_chunks;
        ZK_LAUNCH(k_msm_part_coarse, dim3((unsigned)c_chunks, sh.W), dim3(256), 0, s, wm, sh.n, sh.c, sh.W, c_chunk, sh.sets, level_stride, ptr<u32>(so.off),
                  ptr<u32>(so.ccur), (unsigned long long*)so.pairs.p, keep);
        const u64 max_tiles = total_max / MSM_FINE_TILE + nbins;
        ZK_LAUNCH(k_msm_part_fine, dim3((unsigned)max_tiles), dim3(256), 0, s, (const unsigned long long*)so.pairs.p, ptr<u32>(so.off), ptr<u32>(so.tile_off), nbins,
                  ptr<u32>(so.cursor), ptr<u32>(so.sorted));
    } else {
        ZK_LAUNCH(k_msm_place, dim3((unsigned)sort_chunks, sh.W, halves), dim3(ZK_SORT_THREADS), hist_bytes, s, wm, sh.n, sh.c, sh.W, chunk, sh.sets, kh,
                  level_stride, ptr<u32>(so.off), ptr<u32>(so.cursor), ptr<u32>(so.sorted), keep);
    }
    event_record(so.ready, s);
}

// bucket accumulation + fold for one base table; the bucket sets' weighted sums land in d_window_sums[0..nsums()), two per set.
// Defined in group.cuh and instantiated once per (curve, group) in its own translation unit (bn254_g1.hip, ...):
// the elliptic-curve kernels are by far the most expensive code to compile.
// Runs on lane.stream after so.ready; lane.done is recorded behind the last kernel.
// `d_table` holds packed affine points (AffPacked, level-major when it carries window multiples); the sums come back in
// the saturated Montgomery form.
// `h_window_sums` (may be null): the pinned host mirror of d_window_sums — the lane copies its sums out itself, behind its fold and
// before `done`, so that the host can take each MSM's result as it arrives (Prover::finish) instead of all of them after the last
template <class F>
void msm_run(zkhip_ctx* ctx, MsmLane& lane, const MsmSort& so, const void* d_table, const MsmShape& sh, Xyzz<F>* d_window_sums,
             Event ev_begin, Event ev_end, Event accum_after = nullptr, Xyzz<F>* h_window_sums = nullptr);
template <class F>
void msm_run_tables(zkhip_ctx* ctx, MsmLane& lane, const MsmSort& so, const void* const* d_tables, int nt, const MsmShape& sh, Xyzz<F>* d_window_sums,
                    u32 sum_stride, Event ev_begin, Event ev_end, Event accum_after, Xyzz<F>* h_window_sums = nullptr);
// affine points, saturated Montgomery form -> packed working form of the MSM kernels (level 0 of a table); on ctx->stream
template <class F>
void points_to_packed(zkhip_ctx* ctx, const Aff<F>* d_in, void* d_out, u64 n);
// levels 1 .. L-1 (2^(bits j) P) behind a level 0 of `count` points; synchronises ctx->stream
template <class F>
void msm_table_levels(zkhip_ctx* ctx, void* d_table, u64 count, int bits, int L);
template <class F>
u64 count_infinite(zkhip_ctx* ctx, const void* d_table, u64 count);
template <class F>
void mark_finite(zkhip_ctx* ctx, const void* d_table, u64 count, u32* d_bitmap);
template <class F> static constexpr size_t packed_point_bytes() { return sizeof(AffPacked<typename Unsat<F>::type>); }
// binding a key to a constraint system (bind.cuh: the kernels; group.cuh: these launchers, instantiated for G1 only).  `x`: two
// vectors of N XYZZ points (bind_xyzz_bytes each); everything on ctx->stream.
template <class F>
size_t bind_xyzz_bytes();
template <class F>
void bind_scale(zkhip_ctx* ctx, const void* d_h_table, u64 N, u64 n_src, u32 n1, u32 n2, u32 n3, const u32* d_scal, const u32* d_konst, int nw, void* d_x);
template <class F>
void bind_fft(zkhip_ctx* ctx, void* d_x, u64 N, const u32* d_tw, int nw);
template <class F>
void bind_h_finish(zkhip_ctx* ctx, const void* d_x, u64 N, int logN, void* d_out);
template <class F>
void bind_cmul(zkhip_ctx* ctx, const void* d_x, int logN, const u32* d_row, const u32* d_val, int nw, const u32* d_minus_one, u64 nnz, void* d_prod);
template <class F>
void bind_l_finish(zkhip_ctx* ctx, const void* d_prod, const u64* d_cptr, const void* d_l_table, u64 m, const u32* d_long_cols, u64 n_long, void* d_sum, void* d_out);
// fixed-base tables / multiplications for setup (N3); also per-group code
template <class F>
void fixed_base_table(zkhip_ctx* ctx, const Aff<F>* h_pj, int nwin, DBuf& tbl);
template <class F>
void fixed_base_mul(zkhip_ctx* ctx, const DBuf& tbl, int nwin, const u32* d_scalars, u64 count, Aff<F>* d_out);

// the MSM's value from its bucket-set sums: the single sum of a table MSM, else the host Horner step sum_j 2^(c j) S_j
template <class F>
static Xyzz<F> msm_combine(const Xyzz<F>* ws, const MsmShape& sh) {
    auto set_sum = [&](int j) {                      // one sum per fold digit
        Xyzz<F> t = ws[sh.ndig * j];
        for (u32 d = 1; d < sh.ndig; ++d) t = xyzz_add(t, ws[sh.ndig * j + d]);
        return t;
    };
    if (sh.sets == 1) return set_sum(0);
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int j = (int)sh.sets - 1; j >= 0; --j) {
        for (int i = 0; i < sh.c; ++i) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, set_sum(j));
    }
    return acc;
}

// ------------------------------------------------------------------ byte codecs (host)
template <class F>
static F fe_from_bytes_canon(const uint8_t* b) {   // canonical LE bytes -> canonical limbs (no Montgomery)
    F x;
    memcpy(x.v, b, F::BYTES);
    return x;
}
template <class P>
static bool canon_lt_mod(const Fe<P>& x) {
    for (int i = P::N - 1; i >= 0; --i) {
        if (x.v[i] != P::mod(i)) return x.v[i] < P::mod(i);
    }
    return false;
}
// ark uncompressed affine -> canonical coords with infinity as all-zero (the device sentinel)
template <int FQ_BYTES, int NCOORD>
static void decode_point(const uint8_t* src, uint8_t* dst) {
    constexpr int SZ = FQ_BYTES * NCOORD;
    const bool inf = src[SZ - 1] & 0x40;
    if (inf) { memset(dst, 0, SZ); return; }
    memcpy(dst, src, SZ);
    dst[SZ - 1] &= 0x3f;
}
template <class F> static void write_fe(const F& mont, uint8_t* out) { F c = fe_from_mont(mont); memcpy(out, c.v, F::BYTES); }

}  // namespace zk

// ------------------------------------------------------------------ proving key
struct zkhip_pk {
    int curve;
    int scheme = 0;                               // 0 = Groth16, 1 = GM17 (gm17.cuh: same five MSM lanes, other bases)
    zkhip_ctx* ctx;
    u64 m, w, l, hlen, N;
    int logN;
    DBuf a_ext, b1_ext, l_ext, b2_ext, h_sigma;   // Montgomery affine, MSM-ready
    std::vector<uint8_t> delta_g1_canon;          // for the -rs*delta_1 term of C (host)
    std::vector<uint8_t> g_gamma2_z2_canon;       // GM17: for the rho^2 * g_gamma2_z2 term of C (host)
    // Multi-GPU sharding of ONE proof (zkhip_pk_load_g16_shard): this key holds the bases of the index ranges
    // [z_lo, z_lo + z_n) of the extended variable range [0, m+2) and [h_lo, h_lo + h_n) of the sigma-ordered h range
    // [0, N); every rank derives the same window widths from the nominal (largest) range length.
    u32 rank = 0, world = 1;
    u64 z_lo = 0, z_n = 0, h_lo = 0, h_n = 0;
    int c_z = 0, c_h = 0;
    // The thinned list.  Variables that do not occur in the B matrix have the point at infinity in b_g1_query AND b_g2_query (a
    // third of the Poseidon chain's); a GM17 key holds it for the same half of its variables in a_query, c_query_2 and b_query.
    // The z tables that hold many such bases (a tenth or more) form a family (`thin_mask`: bit k = table k of a_ext, b1_ext,
    // l_ext, b2_ext) whose MSMs pair with a sorted list of their own that leaves out the variables at infinity in ALL of them
    // (`thin_keep`: one bit per entry of the z range, set where some base of the family is finite) — one more counting sort, and
    // the most expensive MSMs of a real circuit shrink by that share.  0: every table on the common list.
    DBuf thin_keep;
    u32 thin_mask = 0;
    bool inf_many_thin[4] = {true, true, true, true};   // inf_many of the family's tables on the thinned list
    bool inf_many[5] = {true, true, true, true, true};   // per table (a, b1, l, b2, h): more than one base in 2048 is the point at
                                                       // infinity -> the accumulation lets lanes sit those out (MsmShape::skip_inf)
    int s_z = 1, s_h = 1;     // bucket sets of the MSMs over z / over h = every s-th window multiple is in the tables (MsmShape::sets)
    // log2 N1 of the NTT plan the sigma order of h_sigma was made for (N = N1 * N2): a context whose plan for this domain
    // splits differently (ZKHIP_TUNE_NTT_SINGLE_MAX_LOG changed after the key was loaded, or an image written under
    // another setting) would pair h with the wrong bases, so the provers and zkhip_pk_import refuse the mismatch
    int ntt_log1 = -1;
    // Bound to one constraint system (zkhip_pk_bind_r1cs, PkLoader::bind): H' = the coset inverse transform applied to h_query
    // (natural order: it pairs with the quotient's EVALUATIONS on the coset), L' = l_query with c's share of the quotient folded in
    // per variable.  Proofs over the system `bound_uid` names take four transforms and two mat-vecs (a GM17 key: two transforms); any
    // other system takes the tables above.  A shard holds its index ranges of H' and L' (zkhip_pk_bind_r1cs_shard).  0 = not bound.
    DBuf h_bound, l_bound;
    u64 bound_uid = 0;
    u64 bound_fp[2] = {0, 0};                // fingerprint of that system (PkLoader::r1cs_fingerprint): what a key image remembers of it — a key
                                             // imported with its bound tables is attached by zkhip_pk_bind_r1cs when the fingerprints agree,
                                             // without recomputing anything
    bool inf_many_bound[2] = {true, true};   // (L', H'): as inf_many
};

// every constraint system of a process has a number of its own (a key remembers WHICH system it was bound to: an address can be
// handed out again after zkhip_r1cs_free)
static inline u64 zk_next_uid() {
    static std::atomic<u64> next{1};
    return next.fetch_add(1);
}
struct zkhip_r1cs {
    u64 uid = zk_next_uid();
    int curve;
    zkhip_ctx* ctx;
    u64 n, l, w, N;
    int logN;
    DBuf rp[3], col[3], val[3];
    u64 nnz[3];
    u64 nnz_short[3];     // without the rows of more than MATVEC_LONG terms (what the lanes-per-row choice of k_matvec is made from)
    DBuf long_rows;       // those rows, matrix << 32 | row (k_matvec_long); the first n_huge of them have more than MATVEC_HUGE terms (k_matvec_huge)
    u64 n_long = 0, n_huge = 0;
    u64 fp[2] = {0, 0};   // PkLoader::r1cs_fingerprint, computed when first asked for
    bool fp_made = false;
};
// the matrices of a resident constraint system back in host memory (setup, N3: key generation walks them on the host; the
// prover never needs them there, so zkhip_r1cs_load keeps no host copy).  Values come back as they are resident:
// saturated Montgomery form.
struct HostCsr {
    std::vector<u64> rp[3];
    std::vector<u32> col[3];
    std::vector<uint8_t> val[3];
    void fetch(zkhip_ctx* ctx, const zkhip_r1cs* cs) {
        for (int k = 0; k < 3; ++k) {
            rp[k].resize(cs->n + 1);
            col[k].resize(cs->nnz[k]);
            val[k].resize(cs->nnz[k] * 32);
            dev_d2h(rp[k].data(), cs->rp[k].p, (cs->n + 1) * 8, ctx->stream);
            if (cs->nnz[k]) {
                dev_d2h(col[k].data(), cs->col[k].p, cs->nnz[k] * 4, ctx->stream);
                dev_d2h(val[k].data(), cs->val[k].p, cs->nnz[k] * 32, ctx->stream);
            }
        }
        stream_sync(ctx->stream);
    }
};

// an assignment resident in HBM: (m + 2) x 32 B canonical integers; the two tail slots receive r and s
struct zkhip_assignment {
    int curve;
    zkhip_ctx* ctx;
    u64 m;
    DBuf scalars;
};

namespace zk {

template <class C>
struct PkLoader {
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    static constexpr int FQB = Fq::BYTES;
    static constexpr int G1B = 2 * FQB, G2B = 4 * FQB;

    struct Rd {
        const uint8_t* p;
        const uint8_t* e;
        const uint8_t* take(size_t n) {
            require((size_t)(e - p) >= n, ZKHIP_ERR_PARSE, "proving key truncated");
            const uint8_t* r = p;
            p += n;
            return r;
        }
        u64 len(size_t elem) {
            u64 n;
            memcpy(&n, take(8), 8);
            require(n <= (u64)(e - p) / elem, ZKHIP_ERR_PARSE, "proving key vector length exceeds file size");
            return n;
        }
    };

    // `count` points of NC base-field coordinates each, resident in Montgomery form: entry shift + j is ark's encoding src[j]
    // (j < nsrc), entry `extra_at` the single point `extra` (if any), every other entry the point at infinity.  Decoded straight
    // into the pinned staging ring on a few host threads (dev_h2d_fill): one pass through host memory.
    template <int NC>
    static void upload_decoded(zkhip_ctx* ctx, DBuf& dst, u64 count, const uint8_t* src, u64 nsrc, u64 shift, const uint8_t* extra, u64 extra_at) {
        constexpr size_t PB = (size_t)FQB * NC;
        dst.ensure(count * PB);
        dev_h2d_fill(dst.p, count * PB, PB, ctx->stream, [=](char* out, size_t off, size_t len) {
            const u64 i0 = off / PB, i1 = (off + len) / PB;
            for (u64 i = i0; i < i1; ++i) {
                uint8_t* o = (uint8_t*)out + (i - i0) * PB;
                if (i >= shift && i - shift < nsrc) decode_point<FQB, NC>(src + (i - shift) * PB, o);
                else if (extra && i == extra_at) decode_point<FQB, NC>(extra, o);
                else memset(o, 0, PB);
            }
        });
        ZK_LAUNCH((k_to_mont<Fq>), dim3(blocks_for(count * NC, 256)), dim3(256), 0, ctx->stream, ptr<Fq>(dst), ptr<Fq>(dst), count * NC);
        stream_sync(ctx->stream);
    }
    // dev[idx] += P (host round trip of one point; P canonical ark encoding)
    template <class F, int NC>
    static void add_into(zkhip_ctx* ctx, DBuf& buf, u64 idx, const uint8_t* ark_point) {
        uint8_t dec[G2B];
        decode_point<FQB, NC>(ark_point, dec);
        Aff<F> add;
        memcpy(&add, dec, sizeof(add));
        add = to_mont_point(add);
        Aff<F> cur;
        dev_d2h(&cur, ptr<Aff<F>>(buf) + idx, sizeof(cur), ctx->stream);
        stream_sync(ctx->stream);
        Aff<F> sum = xyzz_to_affine(xyzz_madd(Xyzz<F>::from_affine(cur), add));
        dev_h2d(ptr<Aff<F>>(buf) + idx, &sum, sizeof(sum), ctx->stream);
        stream_sync(ctx->stream);
    }
    static Aff<Fq> to_mont_point(const Aff<Fq>& p) { return {fe_to_mont(p.x), fe_to_mont(p.y)}; }
    static Aff<Fq2> to_mont_point(const Aff<Fq2>& p) { return {fe_to_mont(p.x), fe_to_mont(p.y)}; }

    static void range_of(u64 total, u32 rank, u32 world, u64& lo, u64& n, u64& nominal) {
        nominal = (total + world - 1) / world;
        lo = std::min<u64>((u64)rank * nominal, total);
        n = std::min<u64>(nominal, total - lo);
    }
    // the layout of ark's `serialize_unchecked` of ark_groth16::ProvingKey
    struct Parsed {
        const uint8_t *alpha_g1, *beta_g2, *delta_g2, *beta_g1, *delta_g1, *a_q, *b1_q, *b2_q, *h_q, *l_q;
        u64 m, w, l, hl, N;
    };
    static Parsed parse(const uint8_t* bytes, size_t len) {
        Rd rd{bytes, bytes + len};
        Parsed k;
        k.alpha_g1 = rd.take(G1B);
        k.beta_g2 = rd.take(G2B);
        rd.take(G2B);                                  // gamma_g2 (verifier only)
        k.delta_g2 = rd.take(G2B);
        const u64 n_abc = rd.len(G1B);
        rd.take(n_abc * G1B);                          // gamma_abc_g1 (verifier only)
        k.beta_g1 = rd.take(G1B);
        k.delta_g1 = rd.take(G1B);
        k.m = rd.len(G1B);
        k.a_q = rd.take(k.m * G1B);
        const u64 mb1 = rd.len(G1B);
        k.b1_q = rd.take(mb1 * G1B);
        const u64 mb2 = rd.len(G2B);
        k.b2_q = rd.take(mb2 * G2B);
        k.hl = rd.len(G1B);
        k.h_q = rd.take(k.hl * G1B);
        k.w = rd.len(G1B);
        k.l_q = rd.take(k.w * G1B);
        require(rd.p == rd.e, ZKHIP_ERR_PARSE, "trailing bytes after proving key");
        require(k.m >= 1 && mb1 == k.m && mb2 == k.m && k.w <= k.m, ZKHIP_ERR_PARSE, "inconsistent query lengths in proving key");
        k.l = k.m - k.w;
        require(n_abc == k.l, ZKHIP_ERR_PARSE, "gamma_abc length != number of instance variables");
        k.N = k.hl + 1;
        require((k.N & (k.N - 1)) == 0, ZKHIP_ERR_PARSE, "h_query length + 1 is not a power of two");
        require(k.m + 2 < ((u64)1 << 31), ZKHIP_ERR_BAD_ARG, "too many variables");
        return k;
    }
    static void load(zkhip_ctx* ctx, const uint8_t* bytes, size_t len, zkhip_pk* pk) {
        const Parsed k = parse(bytes, len);
        const uint8_t *alpha_g1 = k.alpha_g1, *beta_g2 = k.beta_g2, *delta_g2 = k.delta_g2, *beta_g1 = k.beta_g1, *delta_g1 = k.delta_g1, *a_q = k.a_q,
                      *b1_q = k.b1_q, *b2_q = k.b2_q, *h_q = k.h_q, *l_q = k.l_q;
        const u64 m = k.m, w = k.w, l = k.l, hl = k.hl, N = k.N;
        pk->m = m; pk->w = w; pk->l = l; pk->hlen = hl; pk->N = N; pk->logN = ilog2_floor(N);
        NttPlan<C>* plan = get_plan<C>(ctx, pk->logN);
        pk->ntt_log1 = plan->split();
        pk->delta_g1_canon.assign(delta_g1, delta_g1 + G1B);

        const u64 me = m + 2;   // extended by the (delta, r) and (delta, s) pairs — see prove()
        // A_ext = [a_query..., delta_1, inf]          (+ alpha_1 folded into entry 0)
        upload_decoded<2>(ctx, pk->a_ext, me, a_q, m, 0, delta_g1, m);
        // B1_ext = [b_g1_query..., inf, delta_1]      (+ beta_1 folded into entry 0)
        upload_decoded<2>(ctx, pk->b1_ext, me, b1_q, m, 0, delta_g1, m + 1);
        // L_ext = [inf x l, l_query..., inf, inf]
        upload_decoded<2>(ctx, pk->l_ext, me, l_q, w, l, nullptr, 0);
        // B2_ext = [b_g2_query..., inf, delta_2]      (+ beta_2 folded into entry 0)
        upload_decoded<4>(ctx, pk->b2_ext, me, b2_q, m, 0, delta_g2, m + 1);
        // h_query, permuted into the sigma order the NTT pipeline leaves h in, padded with infinity
        DBuf h_nat;
        upload_decoded<2>(ctx, h_nat, std::max<u64>(hl, 1), h_q, hl, 0, nullptr, 0);
        pk->h_sigma.ensure(N * G1B);
        ZK_LAUNCH((k_sigma_gather_points<Aff<Fq>>), dim3(blocks_for(N, 256)), dim3(256), 0, ctx->stream, ptr<Aff<Fq>>(h_nat),
                  ptr<Aff<Fq>>(pk->h_sigma), N, hl, plan->N1, plan->N2, plan->N3);
        stream_sync(ctx->stream);
        h_nat.release();
        // constant terms: z_0 = 1, so alpha/beta ride on entry 0 of their query vectors
        add_into<Fq, 2>(ctx, pk->a_ext, 0, alpha_g1);
        add_into<Fq, 2>(ctx, pk->b1_ext, 0, beta_g1);
        add_into<Fq2, 4>(ctx, pk->b2_ext, 0, beta_g2);
        finish_tables(ctx, pk, me, N);
    }
    // this rank's share of the bases (everything for world = 1) as MSM tables: level 0 = the packed working form of the
    // range's points, levels 1 .. W-1 their window multiples 2^(c j) P (a 2^20 BN254 key: 16 levels, 6 GiB of the 288)
    static void finish_tables(zkhip_ctx* ctx, zkhip_pk* pk, u64 me, u64 hdom) {
        u64 nominal_z, nominal_h;
        range_of(me, pk->rank, pk->world, pk->z_lo, pk->z_n, nominal_z);
        range_of(hdom, pk->rank, pk->world, pk->h_lo, pk->h_n, nominal_h);
        MsmShape shz = msm_shape(ctx, nominal_z, C::Fr::Params::BITS, true), shh = msm_shape(ctx, nominal_h, C::Fr::Params::BITS, true);
        // Every window multiple of every base (one bucket set, one fold per MSM) while that fits the device: a 2^20 key is 6 GiB,
        // 2^24 96 GiB.  Beyond — or when ZKHIP_TUNE_MSM_SETS says so — the tables keep every 2nd, 4th, ... multiple and the MSMs
        // fold as many bucket sets (the reference has no size limit below the field's two-adicity; this is how it is met).
        int sets = ctx->msm_sets;
        if (!sets) {
            const u64 budget = msm_table_budget(ctx, pk->z_n, pk->h_n, hdom, shz.W, shz.K);
            const u64 g1 = packed_point_bytes<Fq>(), g2 = packed_point_bytes<Fq2>();
            sets = 1;
            for (;;) {
                const MsmShape a = msm_shape(ctx, nominal_z, C::Fr::Params::BITS, true, 0, sets), b = msm_shape(ctx, nominal_h, C::Fr::Params::BITS, true, 0, sets);
                const u64 need = pk->z_n * (3 * g1 + g2) * a.levels + pk->h_n * g1 * b.levels;
                if (need <= budget || (int)a.sets < sets) break;      // (a.sets < sets: already one bucket set per window)
                sets *= 2;
            }
        }
        shz = msm_shape(ctx, nominal_z, C::Fr::Params::BITS, true, 0, sets);
        shh = msm_shape(ctx, nominal_h, C::Fr::Params::BITS, true, 0, sets);
        pk->c_z = shz.c;
        pk->c_h = shh.c;
        pk->s_z = (int)shz.sets;
        pk->s_h = (int)shh.sets;
        to_table<Fq>(ctx, pk->a_ext, pk->z_lo, pk->z_n, shz);
        to_table<Fq>(ctx, pk->b1_ext, pk->z_lo, pk->z_n, shz);
        to_table<Fq>(ctx, pk->l_ext, pk->z_lo, pk->z_n, shz);
        to_table<Fq2>(ctx, pk->b2_ext, pk->z_lo, pk->z_n, shz);
        to_table<Fq>(ctx, pk->h_sigma, pk->h_lo, pk->h_n, shh);
        count_points_at_infinity(ctx, pk);
    }
    // levels 1 .. W-1 of the five tables of a key whose level 0 is in place (zkhip_pk_import of a compact image)
    static void table_levels(zkhip_ctx* ctx, zkhip_pk* pk) {
        const MsmShape shz = msm_shape(ctx, pk->z_n, C::Fr::Params::BITS, true, pk->c_z, pk->s_z), shh = msm_shape(ctx, pk->h_n, C::Fr::Params::BITS, true, pk->c_h, pk->s_h);
        msm_table_levels<Fq>(ctx, pk->a_ext.p, pk->z_n, shz.level_bits(), (int)shz.levels);
        msm_table_levels<Fq>(ctx, pk->b1_ext.p, pk->z_n, shz.level_bits(), (int)shz.levels);
        msm_table_levels<Fq>(ctx, pk->l_ext.p, pk->z_n, shz.level_bits(), (int)shz.levels);
        msm_table_levels<Fq2>(ctx, pk->b2_ext.p, pk->z_n, shz.level_bits(), (int)shz.levels);
        msm_table_levels<Fq>(ctx, pk->h_sigma.p, pk->h_n, shh.level_bits(), (int)shh.levels);
        count_points_at_infinity(ctx, pk);
    }
    // ---- zkhip_pk_bind_r1cs: H' and L' for ONE constraint system (bind.cuh has the algebra) ----
    static void unbind(zkhip_pk* pk) {
        pk->bound_uid = 0;
        pk->bound_fp[0] = pk->bound_fp[1] = 0;
        pk->h_bound.release();
        pk->l_bound.release();
    }
    // position-keyed checksum of a resident constraint system (dimensions, the three matrices as they are resident): cached
    static void r1cs_fingerprint(zkhip_ctx* ctx, const zkhip_r1cs* cs_, u64 fp[2]) {
        zkhip_r1cs* cs = const_cast<zkhip_r1cs*>(cs_);
        if (!cs->fp_made) {
            DBuf d;
            d.ensure(16);
            Stream s = ctx->stream;
            dev_memset(d.p, 0, 16, s);
            for (int k = 0; k < 3; ++k) {
                const u64 salt = 0xA5A5A5A5ull * (u64)(k + 1) + cs->n * 0x9E3779B97F4A7C15ull + cs->l * 0xC2B2AE3D27D4EB4Full + cs->w;
                auto run = [&](const void* p, u64 bytes, u64 tag) {
                    if (bytes) ZK_LAUNCH(k_fingerprint, dim3(1024), dim3(256), 0, s, (const u32*)p, bytes / 4, salt ^ (tag * 0x165667B19E3779F9ull), (unsigned long long*)d.p);
                };
                run(cs->rp[k].p, (cs->n + 1) * 8, 1);
                run(cs->col[k].p, cs->nnz[k] * 4, 2);
                run(cs->val[k].p, cs->nnz[k] * 32, 3);
            }
            dev_d2h(cs->fp, d.p, 16, s);
            stream_sync(s);
            cs->fp[0] ^= (u64)cs->curve + 1;           // (never all zero for an empty system either)
            cs->fp_made = true;
        }
        fp[0] = cs->fp[0];
        fp[1] = cs->fp[1];
    }
    // the quotient's domain, the bases it pairs with and the variable range of either scheme
    struct BindShape {
        u64 N, n_src, me, mcols;    // domain; h bases that are not the padding; entries of the l table; variables (columns of W)
        int logN;
    };
    static BindShape bind_shape(const zkhip_pk* pk) {
        BindShape b;
        b.N = pk->N; b.logN = pk->logN; b.mcols = pk->m; b.me = pk->m + 2;
        b.n_src = pk->scheme == 0 ? pk->hlen : pk->N;     // Groth16: h_query has N - 1 entries; GM17: g_gamma2_z_t[0 .. D)
        return b;
    }
    // everything a refusal can depend on, BEFORE anything of the key is touched (zkhip_pk_bind_r1cs)
    static void bind_check(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs) {
        require(pk->curve == C::ID && cs->curve == C::ID, ZKHIP_ERR_BAD_ARG, "curve mismatch between key and constraint system");
        if (pk->scheme == 0) {
            require(pk->m == cs->l + cs->w && pk->w == cs->w && pk->N == cs->N, ZKHIP_ERR_BAD_ARG,
                    "proving key does not match the constraint system (m, w or domain size)");
            require(pk->N >= 2 && pk->hlen + 1 == pk->N, ZKHIP_ERR_BAD_ARG, "domain too small to bind");
        } else {
            const u64 M = 1 + 2 * (cs->l - 1) + cs->w + cs->n, D0 = 2 * cs->n + 2 * (cs->l - 1) + 1;
            require(pk->m == M && pk->l == cs->l && pk->N == ((u64)1 << ilog2_ceil(D0)) && pk->N >= 2, ZKHIP_ERR_BAD_ARG,
                    "GM17 proving key does not match the constraint system (SAP variables, instance size or domain)");
        }
        NttPlan<C>* pl = get_plan<C>(ctx, pk->logN);
        require(pl->split() == pk->ntt_log1, ZKHIP_ERR_BAD_ARG, "the key's h bases were ordered for another NTT split (NTT_SINGLE_MAX_LOG changed): reload the key");
    }
    // W — the matrix whose product with the assignment the quotient subtracts (Groth16: C; GM17: the SAP's) — by COLUMNS, on the
    // host: crow[cptr[v] .. cptr[v+1]) = the rows variable v occurs in, cval the coefficients (saturated Montgomery form, as the
    // matrices are resident), long_cols the variables with more than BIND_SHORT_COL entries
    struct WCols {
        std::vector<u64> cptr;
        std::vector<u32> crow, long_cols;
        std::vector<uint8_t> cval;
        u64 nnz() const { return crow.size(); }
    };
    // `extra[v]`: entries variable v gets besides C's (GM17); each C entry lands on row row_mul * k with its value times val_mul4 ? 4 : 1
    struct WExtra { u32 col, row; int four; };
    static void w_columns(zkhip_ctx* ctx, const zkhip_r1cs* cs, u64 mcols, u32 row_mul, bool val_mul4, const std::vector<WExtra>& extra, WCols& W) {
        typedef typename C::Fr Fr;
        static_assert(sizeof(Fr) == 32 && Fr::N <= 12, "the binding's scalar routines take up to 12 words; Fr is 32 bytes on both curves");
        Stream s = ctx->stream;
        const u64 nnz = cs->nnz[2];
        std::vector<u64> rp(cs->n + 1);
        std::vector<u32> col(nnz);
        std::vector<uint8_t> val(nnz * sizeof(Fr));
        dev_d2h(rp.data(), cs->rp[2].p, (cs->n + 1) * 8, s);
        if (nnz) {
            dev_d2h(col.data(), cs->col[2].p, nnz * 4, s);
            dev_d2h(val.data(), cs->val[2].p, nnz * sizeof(Fr), s);
        }
        stream_sync(s);
        W.cptr.assign(mcols + 1, 0);
        for (u64 e = 0; e < nnz; ++e) {
            require(col[e] < mcols, ZKHIP_ERR_BAD_ARG, "internal: column index out of range");
            ++W.cptr[col[e] + 1];
        }
        for (const WExtra& x : extra) ++W.cptr[x.col + 1];
        for (u64 v = 0; v < mcols; ++v) W.cptr[v + 1] += W.cptr[v];
        const u64 total = W.cptr[mcols];
        W.crow.resize(total);
        W.cval.resize(total * sizeof(Fr));
        std::vector<u64> cur(W.cptr.begin(), W.cptr.end() - 1);
        for (u64 k = 0; k < cs->n; ++k)
            for (u64 e = rp[k]; e < rp[k + 1]; ++e) {
                const u64 at = cur[col[e]]++;
                W.crow[at] = (u32)(k * row_mul);
                Fr v;
                memcpy(v.v, &val[e * sizeof(Fr)], sizeof(Fr));
                if (val_mul4) v = fe_dbl(fe_dbl(v));
                memcpy(&W.cval[at * sizeof(Fr)], v.v, sizeof(Fr));
            }
        const Fr one = Fr::one(), four = fe_dbl(fe_dbl(Fr::one()));
        for (const WExtra& x : extra) {
            const u64 at = cur[x.col]++;
            W.crow[at] = x.row;
            memcpy(&W.cval[at * sizeof(Fr)], (x.four ? four : one).v, sizeof(Fr));
        }
        W.long_cols.clear();
        for (u64 v = 0; v < mcols; ++v)
            if (W.cptr[v + 1] - W.cptr[v] > BIND_SHORT_COL) W.long_cols.push_back((u32)v);
    }
    // Level 0 of H' (N points, NATURAL order) and of L' (me points) from level 0 of the key's h table (sigma order, N entries, packed)
    // and of its padded l table (me entries, packed), both covering the WHOLE index range, on this context's device.
    static void bind_level0(zkhip_ctx* ctx, NttPlan<C>* pl, const BindShape& b, const void* d_h0, const void* d_l0, const WCols& W, DBuf& h_out, DBuf& l_out) {
        typedef typename C::Fr Fr;
        constexpr int NW = Fr::N;
        const u64 N = b.N, nnz = W.nnz();
        const u64 g1 = packed_point_bytes<Fq>(), xb = bind_xyzz_bytes<Fq>();
        {   // the transforms' points and W's products are transient, like the levels' workspace
            size_t free_b = 0, total_b = 0;
            dev_mem_info(&free_b, &total_b);
            const u64 need = (N + b.me) * g1 + 2 * N * xb + nnz * (xb + 4 + 32) + b.mcols * (xb + 8) + N * 48 + ((u64)1 << 30);
            require(need < free_b, ZKHIP_ERR_NOMEM, "not enough device memory to bind the key (the transforms' workspace)");
        }
        Stream s = ctx->stream;
        const unsigned T = 256;
        // canonical-integer scalars: g^-i / N (what the bases of H' are scaled by), w^-e (the transform's roots), -1 / (N Z(g)), p - 1
        DBuf d_scal, d_tw, d_k;
        d_scal.ensure(N * sizeof(Fr));
        d_tw.ensure(std::max<u64>(N / 2, 1) * sizeof(Fr));
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(N, T)), dim3(T), 0, s, ptr<Fr>(d_scal), pl->g_inv, pl->n_inv, N, 0u, 0u, 0, 1u);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(N, T)), dim3(T), 0, s, ptr<Fr>(d_scal), ptr<Fr>(d_scal), N);
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(N / 2, T)), dim3(T), 0, s, ptr<Fr>(d_tw), pl->omega_inv, Fr::one(), N / 2, 0u, 0u, 0, 1u);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(N / 2, T)), dim3(T), 0, s, ptr<Fr>(d_tw), ptr<Fr>(d_tw), N / 2);
        Fr konst[2];
        konst[0] = fe_from_mont(fe_neg(fe_mul(pl->n_inv, pl->zinv)));
        konst[1] = fe_from_mont(fe_neg(Fr::one()));
        d_k.ensure(sizeof(konst));
        dev_h2d(d_k.p, konst, sizeof(konst), s);
        stream_sync(s);                                // (konst is on this frame)
        // the two transforms over the bases: vector 0 -> H' (coset), vector 1 -> H'' (what W's columns are summed against)
        DBuf x;
        x.ensure(2 * N * xb);
        bind_scale<Fq>(ctx, d_h0, N, b.n_src, pl->N1, pl->N2, pl->N3, ptr<u32>(d_scal), ptr<u32>(d_k), NW, x.p);
        bind_fft<Fq>(ctx, x.p, N, ptr<u32>(d_tw), NW);
        h_out.ensure(N * g1);
        bind_h_finish<Fq>(ctx, x.p, N, b.logN, h_out.p);
        stream_sync(s);
        d_scal.release();
        d_tw.release();
        // W by columns: D_v = sum_k W[k][v] H''_k, added to the l base of v
        DBuf d_cptr, d_crow, d_cval, d_long, prod, sum;
        d_cptr.ensure((b.mcols + 1) * 8);
        d_crow.ensure(std::max<u64>(nnz, 1) * 4);
        d_cval.ensure(std::max<u64>(nnz, 1) * 32);
        d_long.ensure(std::max<u64>(W.long_cols.size(), 1) * 4);
        prod.ensure(std::max<u64>(nnz, 1) * xb);
        sum.ensure(std::max<u64>(b.mcols, 1) * xb);
        dev_h2d(d_cptr.p, W.cptr.data(), (b.mcols + 1) * 8, s);
        if (!W.long_cols.empty()) dev_h2d(d_long.p, W.long_cols.data(), W.long_cols.size() * 4, s);
        if (nnz) {
            dev_h2d(d_crow.p, W.crow.data(), nnz * 4, s);
            dev_h2d(d_cval.p, W.cval.data(), nnz * 32, s);
            ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(nnz, T)), dim3(T), 0, s, ptr<Fr>(d_cval), ptr<Fr>(d_cval), nnz);   // resident values: Montgomery form
            bind_cmul<Fq>(ctx, (const uint8_t*)x.p + N * xb, b.logN, ptr<u32>(d_crow), ptr<u32>(d_cval), NW, ptr<u32>(d_k) + NW, nnz, prod.p);
        }
        l_out.ensure(b.me * g1);
        dev_d2d((uint8_t*)l_out.p + b.mcols * g1, (const uint8_t*)d_l0 + b.mcols * g1, (b.me - b.mcols) * g1, s);   // the slots behind the variables: as in l's table
        bind_l_finish<Fq>(ctx, prod.p, ptr<u64>(d_cptr), d_l0, b.mcols, ptr<u32>(d_long), W.long_cols.size(), sum.p, l_out.p);
        stream_sync(s);                                // (the host vectors were the copies' sources)
    }
    // this key's index ranges of H' / L' as MSM tables (level 0 from `h_src` / `l_src`: this key's RANGES, h_n / z_n points, on the
    // host or — `on_device` — on this device), the window multiples behind them, the counts the accumulation kernel is chosen from
    static void install_bound(zkhip_ctx* ctx, zkhip_pk* pk, const void* h_src, const void* l_src, bool on_device, const u64 fp[2]) {
        typedef typename C::Fr Fr;
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const u64 g1 = packed_point_bytes<Fq>();
        {
            size_t free_b = 0, total_b = 0;
            dev_mem_info(&free_b, &total_b);
            const u64 need = pk->h_n * shh.levels * g1 + pk->z_n * shz.levels * g1 + ((u64)2 << 30);
            require(need < free_b, ZKHIP_ERR_NOMEM, "not enough device memory to bind the key (two more MSM tables)");
        }
        Stream s = ctx->stream;
        pk->h_bound.ensure(std::max<u64>(pk->h_n, 1) * (u64)shh.levels * g1);
        pk->l_bound.ensure(std::max<u64>(pk->z_n, 1) * (u64)shz.levels * g1);
        auto put = [&](DBuf& dst, const void* src, u64 bytes) {
            if (!bytes) return;
            if (on_device) dev_d2d(dst.p, src, bytes, s);
            else dev_h2d_fill(dst.p, bytes, 64, s, [src](char* out, size_t off, size_t len) { memcpy(out, (const char*)src + off, len); });
        };
        put(pk->h_bound, h_src, pk->h_n * g1);
        put(pk->l_bound, l_src, pk->z_n * g1);
        stream_sync(s);
        msm_table_levels<Fq>(ctx, pk->h_bound.p, pk->h_n, shh.level_bits(), (int)shh.levels);
        msm_table_levels<Fq>(ctx, pk->l_bound.p, pk->z_n, shz.level_bits(), (int)shz.levels);
        pk->inf_many_bound[0] = count_infinite<Fq>(ctx, pk->l_bound.p, pk->z_n) * 2048 > pk->z_n;
        pk->inf_many_bound[1] = count_infinite<Fq>(ctx, pk->h_bound.p, pk->h_n) * 2048 > pk->h_n;
        pk->bound_fp[0] = fp[0];
        pk->bound_fp[1] = fp[1];
    }
    static void w_columns_for(zkhip_ctx* ctx, int scheme, const zkhip_r1cs* cs, u64 mcols, WCols& W) {
        if (scheme == 0) { w_columns(ctx, cs, mcols, 1, false, {}, W); return; }
        // GM17: W = the SAP's right-hand sides (gm17.cuh: rows 2k: 4 C_k + e_k, 2k+1: e_k, 2n: 1, 2n+2i-1: 4 x_i + f_i, 2n+2i: f_i)
        const u64 n = cs->n, l = cs->l, m = cs->l + cs->w;
        std::vector<WExtra> extra;
        extra.reserve(2 * n + 3 * l);
        extra.push_back({0u, (u32)(2 * n), 0});
        for (u64 i = 1; i < l; ++i) extra.push_back({(u32)i, (u32)(2 * n + 2 * i - 1), 1});
        for (u64 k = 0; k < n; ++k) { extra.push_back({(u32)(m + k), (u32)(2 * k), 0}); extra.push_back({(u32)(m + k), (u32)(2 * k + 1), 0}); }
        for (u64 i = 1; i < l; ++i) { extra.push_back({(u32)(m + n - 1 + i), (u32)(2 * n + 2 * i - 1), 0}); extra.push_back({(u32)(m + n - 1 + i), (u32)(2 * n + 2 * i), 0}); }
        w_columns(ctx, cs, mcols, 2, true, extra, W);
    }
    // zkhip_pk_bind_r1cs on a key that covers the whole index range (or one whose imported bound tables only need attaching)
    static void bind(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* cs) {
        bind_check(ctx, pk, cs);
        u64 fp[2];
        r1cs_fingerprint(ctx, cs, fp);
        if (pk->h_bound.p && pk->l_bound.p && pk->bound_fp[0] == fp[0] && pk->bound_fp[1] == fp[1] && (fp[0] | fp[1])) {
            pk->bound_uid = cs->uid;                    // the tables are there already (an imported image, or the same system loaded again)
            return;
        }
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG,
                "a shard of a multi-GPU key binds through zkhip_pk_bind_r1cs_shard / zkhip_multi_bind (the transforms need every base of the key once)");
        unbind(pk);
        NttPlan<C>* pl = get_plan<C>(ctx, pk->logN);
        const BindShape b = bind_shape(pk);
        require(pk->z_n == b.me && pk->h_n == b.N, ZKHIP_ERR_BAD_ARG, "internal: an unsharded key covers the whole index range");
        WCols W;
        w_columns_for(ctx, pk->scheme, cs, b.mcols, W);
        DBuf h0, l0;
        bind_level0(ctx, pl, b, pk->h_sigma.p, pk->l_ext.p, W, h0, l0);
        install_bound(ctx, pk, h0.p, l0.p, true, fp);
        pk->bound_uid = cs->uid;
    }
    // Level 0 of H' and L' over the WHOLE index range, from the key FILE (a shard holds only its ranges of the bases; the transforms
    // need all of them once), left in host memory: every member of a multi-GPU prover then installs its own ranges
    // (zkhip_multi_bind: one member computes, all install; zkhip_pk_bind_r1cs_shard: a rank of a multi-process prover does both).
    static void bound_level0_from_file(zkhip_ctx* ctx, int scheme, const zkhip_r1cs* cs, const uint8_t* bytes, size_t len, std::vector<uint8_t>& h_host,
                                       std::vector<uint8_t>& l_host, u64 fp[2]) {
        zkhip_pk tmp;                                   // dimensions and level 0 of the two tables only
        tmp.curve = C::ID; tmp.scheme = scheme; tmp.ctx = ctx;
        DBuf h_aff, l_aff;
        if (scheme == 0) {
            const Parsed k = parse(bytes, len);
            tmp.m = k.m; tmp.w = k.w; tmp.l = k.l; tmp.hlen = k.hl; tmp.N = k.N; tmp.logN = ilog2_floor(k.N);
            upload_decoded<2>(ctx, l_aff, k.m + 2, k.l_q, k.w, k.l, nullptr, 0);
            upload_decoded<2>(ctx, h_aff, std::max<u64>(k.hl, 1), k.h_q, k.hl, 0, nullptr, 0);
        } else {
            gm17_level0_sources(ctx, bytes, len, &tmp, h_aff, l_aff);
        }
        NttPlan<C>* pl = get_plan<C>(ctx, tmp.logN);
        tmp.ntt_log1 = pl->split();
        bind_check(ctx, &tmp, cs);
        const BindShape b = bind_shape(&tmp);
        const u64 g1 = packed_point_bytes<Fq>();
        DBuf h_sig, h0p, l0p;
        h_sig.ensure(b.N * G1B);
        ZK_LAUNCH((k_sigma_gather_points<Aff<Fq>>), dim3(blocks_for(b.N, 256)), dim3(256), 0, ctx->stream, ptr<Aff<Fq>>(h_aff), ptr<Aff<Fq>>(h_sig), b.N, b.n_src,
                  pl->N1, pl->N2, pl->N3);
        h0p.ensure(b.N * g1);
        l0p.ensure(b.me * g1);
        points_to_packed<Fq>(ctx, ptr<Aff<Fq>>(h_sig), h0p.p, b.N);
        points_to_packed<Fq>(ctx, ptr<Aff<Fq>>(l_aff), l0p.p, b.me);
        stream_sync(ctx->stream);
        h_aff.release(); l_aff.release(); h_sig.release();
        WCols W;
        w_columns_for(ctx, scheme, cs, b.mcols, W);
        DBuf h0, l0;
        bind_level0(ctx, pl, b, h0p.p, l0p.p, W, h0, l0);
        h_host.resize(b.N * g1);
        l_host.resize(b.me * g1);
        dev_d2h(h_host.data(), h0.p, h_host.size(), ctx->stream);
        dev_d2h(l_host.data(), l0.p, l_host.size(), ctx->stream);
        stream_sync(ctx->stream);
        r1cs_fingerprint(ctx, cs, fp);
    }
    // ark_gm17::ProvingKey: the quotient's bases g_gamma2_z_t[0 .. D) and the padded c_query_1 table (gm17.cuh Gm17::load's lane 2)
    static void gm17_level0_sources(zkhip_ctx* ctx, const uint8_t* bytes, size_t len, zkhip_pk* dims, DBuf& h_aff, DBuf& l_aff) {
        Rd rd{bytes, bytes + len};
        rd.take(G2B); rd.take(G1B); rd.take(G2B); rd.take(G1B); rd.take(G2B);
        const u64 l = rd.len(G1B);
        rd.take(l * G1B);
        const u64 M = rd.len(G1B);
        rd.take(M * G1B);
        const u64 Mb = rd.len(G2B);
        rd.take(Mb * G2B);
        const u64 n1 = rd.len(G1B);
        const uint8_t* c1_q = rd.take(n1 * G1B);
        const u64 Mc2 = rd.len(G1B);
        rd.take(Mc2 * G1B);
        rd.take(G1B); rd.take(G2B);
        const uint8_t* g_ab_gamma_z = rd.take(G1B);
        rd.take(G1B);
        const u64 tl = rd.len(G1B);
        const uint8_t* t_q = rd.take(tl * G1B);
        require(rd.p == rd.e, ZKHIP_ERR_PARSE, "trailing bytes after proving key");
        require(l >= 1 && M >= l && Mb == M && Mc2 == M && n1 == M - l && tl >= 2 && ((tl - 1) & (tl - 2)) == 0 && M + 2 < ((u64)1 << 31), ZKHIP_ERR_PARSE,
                "inconsistent query lengths in GM17 proving key");
        const u64 D = tl - 1;
        dims->m = M; dims->w = n1; dims->l = l; dims->hlen = tl; dims->N = D; dims->logN = ilog2_floor(D);
        upload_decoded<2>(ctx, l_aff, M + 2, c1_q, n1, l, g_ab_gamma_z, M);
        upload_decoded<2>(ctx, h_aff, D, t_q, D, 0, nullptr, 0);
    }
    // a member's share of a binding computed elsewhere: its ranges of the whole-range level-0 arrays
    static void install_bound_ranges(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* h_full, size_t h_len, const uint8_t* l_full, size_t l_len,
                                     const u64 fp[2]) {
        bind_check(ctx, pk, cs);
        const BindShape b = bind_shape(pk);
        const u64 g1 = packed_point_bytes<Fq>();
        require(h_len == b.N * g1 && l_len == b.me * g1, ZKHIP_ERR_BAD_ARG, "internal: bound level-0 arrays of another key");
        u64 have[2];
        r1cs_fingerprint(ctx, cs, have);
        require(have[0] == fp[0] && have[1] == fp[1], ZKHIP_ERR_BAD_ARG, "internal: the binding was computed for another constraint system");
        unbind(pk);
        install_bound(ctx, pk, h_full + pk->h_lo * g1, l_full + pk->z_lo * g1, false, fp);
        pk->bound_uid = cs->uid;
    }
    static void count_points_at_infinity(zkhip_ctx* ctx, zkhip_pk* pk) {
        const u64 cnt[5] = {count_infinite<Fq>(ctx, pk->a_ext.p, pk->z_n), count_infinite<Fq>(ctx, pk->b1_ext.p, pk->z_n), count_infinite<Fq>(ctx, pk->l_ext.p, pk->z_n),
                            count_infinite<Fq2>(ctx, pk->b2_ext.p, pk->z_n), count_infinite<Fq>(ctx, pk->h_sigma.p, pk->h_n)};
        for (int k = 0; k < 5; ++k) pk->inf_many[k] = cnt[k] * 2048 > (k == 4 ? pk->h_n : pk->z_n);
        // the thinned list (zkhip_pk::thin_mask): ZKHIP_TUNE_B_SORT 0 = the tables a tenth of whose bases are at infinity, if leaving
        // out what they share drops a tenth of the list; 1 = the families the key formats predict, whatever the counts; 2 = never
        const bool gm17 = pk->scheme == 1;
        u32 family = 0;
        if (ctx->b_sort_mode == 1) family = gm17 ? 0xbu : 0xau;          // GM17: a, c_query_2, b; Groth16: b_g1, b_g2
        else if (ctx->b_sort_mode == 0)
            for (int k = 0; k < 4; ++k)
                if (cnt[k] * 10 > pk->z_n) family |= 1u << k;
        pk->thin_mask = 0;
        pk->thin_keep.release();
        if (family && pk->z_n) {
            const size_t words = (size_t)((pk->z_n + 31) / 32);
            pk->thin_keep.ensure(words * 4);
            dev_memset(pk->thin_keep.p, 0, words * 4, ctx->stream);
            if (family & 1) mark_finite<Fq>(ctx, pk->a_ext.p, pk->z_n, ptr<u32>(pk->thin_keep));
            if (family & 2) mark_finite<Fq>(ctx, pk->b1_ext.p, pk->z_n, ptr<u32>(pk->thin_keep));
            if (family & 4) mark_finite<Fq>(ctx, pk->l_ext.p, pk->z_n, ptr<u32>(pk->thin_keep));
            if (family & 8) mark_finite<Fq2>(ctx, pk->b2_ext.p, pk->z_n, ptr<u32>(pk->thin_keep));
            std::vector<u32> host(words);
            dev_d2h(host.data(), pk->thin_keep.p, words * 4, ctx->stream);
            stream_sync(ctx->stream);
            u64 left_out = 0;
            for (u64 i = 0; i < pk->z_n; ++i) left_out += !((host[i >> 5] >> (i & 31)) & 1u);
            if (ctx->b_sort_mode == 1 || left_out * 10 > pk->z_n) {
                pk->thin_mask = family;
                // what the family's tables still meet at infinity on the thinned list
                for (int k = 0; k < 4; ++k) pk->inf_many_thin[k] = (cnt[k] - std::min(cnt[k], left_out)) * 2048 > pk->z_n;
            } else {
                pk->thin_keep.release();          // (the tables' points at infinity do not coincide: nothing to gain)
            }
        }
    }
    template <class F>
    static void to_table(zkhip_ctx* ctx, DBuf& buf, u64 lo, u64 count, const MsmShape& sh) {
        DBuf out;
        out.ensure(std::max<u64>(count, 1) * (u64)sh.levels * packed_point_bytes<F>());
        if (count) points_to_packed<F>(ctx, ptr<Aff<F>>(buf) + lo, out.p, count);
        stream_sync(ctx->stream);
        buf.swap(out);
        out.release();                        // the saturated copy goes before the levels' workspace is allocated
        msm_table_levels<F>(ctx, buf.p, count, sh.level_bits(), (int)sh.levels);
    }
};

// ------------------------------------------------------------------ prover
template <class C>
struct Prover {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    static constexpr int FQB = Fq::BYTES;

    static CsrDev csr(const zkhip_r1cs* cs, int k) { return CsrDev{ptr<u64>(cs->rp[k]), ptr<u32>(cs->col[k]), cs->val[k].p}; }

    // K1: a = A z, b = B z, c = C z over rows [0, n) (+ the l instance rows of A), zero-filled up to N
    // (`nmat` = 2: A and B only — a key bound to the system carries c's share in its bases; the long rows of C, if any, still
    // land in the c vector, which nothing reads then)
    // (`mat0`: the first matrix of the launch — a member of a multi-GPU proof that transforms b only runs mat0 = 1, nmat = 1)
    static void matvec(zkhip_ctx* ctx, const zkhip_r1cs* cs, const Fr* zmont, Fr* a, Fr* b, Fr* c, u64 n, u64 l, u64 N, int nmat = 3, int mat0 = 0) {
        int g[3];
        for (int k = 0; k < 3; ++k) g[k] = (k >= mat0 && k < mat0 + nmat) ? matvec_group(cs->nnz_short[k], cs->n) : 1;
        ZK_LAUNCH((k_matvec<Fr>), dim3(blocks_for(N, 256 / gmax_rows(g)), nmat), dim3(256), 0, ctx->ws, csr(cs, 0), csr(cs, 1), csr(cs, 2), zmont, a, b, c, n,
                  l, N, g[0], g[1], g[2], cs->l + cs->w, mat0);
        if (cs->n_huge)
            ZK_LAUNCH((k_matvec_huge<Fr>), dim3((unsigned)cs->n_huge), dim3(MATVEC_HUGE_THREADS), 0, ctx->ws, csr(cs, 0), csr(cs, 1), csr(cs, 2), zmont, a, b, c,
                      ptr<u64>(cs->long_rows), cs->l + cs->w);
        if (cs->n_long > cs->n_huge)
            ZK_LAUNCH((k_matvec_long<Fr>), dim3(blocks_for(cs->n_long - cs->n_huge, 4)), dim3(256), 0, ctx->ws, csr(cs, 0), csr(cs, 1), csr(cs, 2), zmont, a, b, c,
                      ptr<u64>(cs->long_rows) + cs->n_huge, cs->n_long - cs->n_huge, cs->l + cs->w);
    }
    static unsigned gmax_rows(const int g[3]) { return (unsigned)std::max(g[0], std::max(g[1], g[2])); }

    // K1-K4 on the device: leaves h (canonical integers, sigma order) in ctx->cur->va.  The three vectors a, b, c live
    // back to back in va and go through every pass together (one launch per pass, grid.y = 3).
    // `bound`: the key is bound to this system (PkLoader::bind) — the transforms that lead from the quotient's evaluations to h's
    // coefficients, and everything c needs, were applied to the key's bases once: what is left per proof is a and b to their
    // coefficients and on to the coset (FOUR transforms), and U_j = a_j b_j / Z(g) as canonical integers in NATURAL order in va.
    static void witness_map(zkhip_ctx* ctx, const zkhip_r1cs* cs, NttPlan<C>* pl, bool bound = false, int half = -1) {
        Stream s = ctx->ws;
        const u64 N = pl->N;
        ctx->cur->va.ensure(3 * N * sizeof(Fr));
        Fr *a = ptr<Fr>(ctx->cur->va), *b = a + N, *c = b + N;
        if (half >= 0) {
            // ONE of the two vectors a bound key's proof needs on the coset (a member of a multi-GPU proof: its partner of the other
            // parity computes the other, Prover::enqueue_tail multiplies them once both are here): two transforms instead of four
            Fr* v = a + (u64)half * N;
            matvec(ctx, cs, ptr<Fr>(ctx->cur->zmont), a, b, c, cs->n, cs->l, N, 1, half);
            event_record(ctx->cur->ntt_b, s);
            ntt_kind_a<C>(ctx, pl, v, true, ptr<Fr>(pl->s_coset), 1, 0);
            ntt_kind_b<C>(ctx, pl, v, false, nullptr, 1, 0);
            event_record(ctx->cur->ntt_e, s);
            return;
        }
        matvec(ctx, cs, ptr<Fr>(ctx->cur->zmont), a, b, c, cs->n, cs->l, N, bound ? 2 : 3);
        event_record(ctx->cur->ntt_b, s);
        if (bound) {
            ntt_kind_a<C>(ctx, pl, a, true, ptr<Fr>(pl->s_coset), 2, N);    // a, b: ifft, then * g^i
            ntt_kind_b<C>(ctx, pl, a, false, nullptr, 2, N);                 // a, b: evaluations on g<w>, natural order
            // (a plain-integer factor takes an R'-form product out of the Montgomery domain: canonical integers, the MSM's digits)
            ZK_LAUNCH((k_quotient<typename Fr::Params>), dim3(blocks_for(N, 256)), dim3(256), 0, s, a, b, fe_from_mont(pl->zinv), a, N);
            event_record(ctx->cur->ntt_e, s);
            return;
        }
        // SIX transforms (the reference's witness_map runs seven): a and b go to the coset and back as a product; c only needs
        // its coefficients — ((ab - c)/Z)'s coefficients are coset_ifft(ab / Z) - c_coeffs / Z, because the coset transform pair is
        // the identity on the c term — so c takes ONE inverse transform whose exit factor carries 1/(N Z) and leaves canonical
        // integers, and the last transform subtracts them where it stores h.  Same h, bit for bit.
        ntt_kind_a<C>(ctx, pl, a, true, ptr<Fr>(pl->s_coset), 2, N);    // a, b: ifft, then * g^i   (coset shift)
        ntt_kind_a<C>(ctx, pl, c, true, ptr<Fr>(pl->s_cexit), 1, 0, 1);  // c: ifft * 1/(N Z), canonical integers, sigma order
        ntt_kind_b<C>(ctx, pl, a, false, nullptr, 2, N);                 // a, b: evaluations on g<w>
        ZK_LAUNCH((k_quotient<typename Fr::Params>), dim3(blocks_for(N, 256)), dim3(256), 0, s, a, b, pl->zinv_rp, a, N);
        ntt_kind_a<C>(ctx, pl, a, true, ptr<Fr>(pl->s_cosetinv_canon), 1, 0, 1, c);   // coset_ifft, canonical, minus c's share
        event_record(ctx->cur->ntt_e, s);
    }

    // z -> HBM (canonical integers; slots m, m+1 are reserved for r, s)
    static void upload_z(zkhip_ctx* ctx, DBuf& dst, u64 m, const uint8_t* z, DBuf& flag) {
        Fr z0 = fe_from_bytes_canon<Fr>(z);
        Fr one = Fr::zero(); one.v[0] = 1;
        require(z0.equals(one), ZKHIP_ERR_BAD_ARG, "z[0] must be 1 (ark instance variable 0 is the constant ONE)");
        dst.ensure((m + 2) * 32);
        dev_h2d(dst.p, z, m * 32, ctx->stream);
        // every entry must be a canonical field element (checked on the device; the verdict is read with the results)
        flag.ensure(4);
        dev_memset(flag.p, 0, 4, ctx->stream);
        ZK_LAUNCH((k_check_canonical<Fr>), dim3(blocks_for(m, 256)), dim3(256), 0, ctx->stream, ptr<Fr>(dst), m, ptr<u32>(flag));
    }
    static void require_canonical(u32 flag) {
        require(flag == 0, ZKHIP_ERR_BAD_ARG, "an assignment entry is not a canonical field element (>= r)");
    }
    // r, s into the tail slots; Montgomery copy of z for the mat-vec
    static void stage_scalars(zkhip_ctx* ctx, NttPlan<C>* pl, void* d_scalars, u64 m, const uint8_t* r, const uint8_t* s_) {
        Stream s = ctx->stream;
        ctx->cur->zmont.ensure(m * 32);
        dev_h2d((uint8_t*)d_scalars + m * 32, r, 32, s);
        dev_h2d((uint8_t*)d_scalars + (m + 1) * 32, s_, 32, s);
        // z in R'-form for the mat-vec (the matrices' values stay in the saturated Montgomery form: their product is R'-form)
        ZK_LAUNCH((k_mul_const<Fr>), dim3(blocks_for(m, 256)), dim3(256), 0, s, (const Fr*)d_scalars, ptr<Fr>(ctx->cur->zmont), m, pl->k_to_rp);
    }

    // ---- enqueue: every kernel and copy of one proof, no host synchronisation
    // src_dev != nullptr: the assignment is already in HBM (copied device-to-device into the slot); else z is a host buffer
    // `lone`: nothing else of this context is in flight beside this proof (the single-proof entry points; a batch pipelines)
    static void enqueue(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* src_dev,
                        const uint8_t* r, const uint8_t* s_, bool lone = false) {
        enqueue_head(ctx, sl, pk, cs, z_host, src_dev, r, s_, lone, -1);
        enqueue_tail(ctx, sl, pk, cs);
    }
    // whether a proof over (pk, cs) may be split between members: a bound key (its proof needs a and b on the coset and nothing else
    // of the witness map) over a domain where two transforms cost more than the exchange
    static bool can_split(const zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs) {
        return pk->scheme == 0 && pk->bound_uid != 0 && pk->bound_uid == cs->uid && pk->logN >= ctx->split_min_log;
    }
    // the staged scalars, the sort of the assignment, the G2 lane and the witness map — all of it (`half` = -1) or the vector `half`
    // (0: a, 1: b) of a bound key's, after which sl.half_ready is recorded and the caller brings the other vector into
    // sl.va + (1 - half) * N before enqueue_tail
    static void enqueue_head(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* src_dev,
                             const uint8_t* r, const uint8_t* s_, bool lone, int half) {
        require(pk->curve == C::ID && cs->curve == C::ID, ZKHIP_ERR_BAD_ARG, "curve mismatch between key and constraint system");
        require(pk->scheme == 0, ZKHIP_ERR_BAD_ARG, "this is a GM17 proving key: use zkhip_prove_gm17");
        require(pk->m == cs->l + cs->w && pk->w == cs->w && pk->N == cs->N, ZKHIP_ERR_BAD_ARG,
                "proving key does not match the constraint system (m, w or domain size)");
        require(!sl.busy, ZKHIP_ERR_DEVICE, "internal: proof slot still in flight");
        make_pipe_streams(ctx);     // (a resident prover's first proof: every stream of the plan, in pipe order)
        slot_init(ctx, sl);
        const u64 m = pk->m, N = pk->N;
        const bool bound = pk->bound_uid != 0 && pk->bound_uid == cs->uid;   // (zkhip_pk_bind_r1cs; a shard: its ranges of H' / L')
        require(half < 0 || (bound && half <= 1), ZKHIP_ERR_BAD_ARG, "internal: only a proof over a bound key splits its witness map");
        sl.half = half;
        sl.lone = lone;
        Fr rr = fe_from_bytes_canon<Fr>(r), ss = fe_from_bytes_canon<Fr>(s_);
        require(canon_lt_mod(rr) && canon_lt_mod(ss), ZKHIP_ERR_BAD_ARG, "r or s not a canonical field element");
        NttPlan<C>* pl = get_plan<C>(ctx, pk->logN);
        require(pl->split() == pk->ntt_log1, ZKHIP_ERR_BAD_ARG, "the key's h bases were ordered for another NTT split (NTT_SINGLE_MAX_LOG changed): reload the key");
        sl.t_start = std::chrono::steady_clock::now();
        memcpy(sl.r, r, 32);
        memcpy(sl.s, s_, 32);
        ctx->cur = &sl;
        Stream st = ctx->stream;
        ctx->ws = st;
        if (z_host) {
            upload_z(ctx, sl.scalars, m, z_host, sl.zflag);
        } else {
            sl.scalars.ensure((m + 2) * 32);
            dev_d2d(sl.scalars.p, src_dev, m * 32, st);
            sl.zflag.ensure(4);
            dev_memset(sl.zflag.p, 0, 4, st);      // a resident assignment was checked when it was uploaded
        }
        void* d_scalars = sl.scalars.p;
        stage_scalars(ctx, pl, d_scalars, m, r, s_);
        event_record(sl.ev[0], st);

        // ---- MSMs over S = [z_0..z_{m-1}, r, s]: A, B1, L in G1 and B2 in G2 share one digit/sort pass
        // (a sharded key covers only its index range of the bases, and pairs them with the same range of the scalars)
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        // (the shape is recomputed from the key's own (c, sets): a key whose numbers this build cannot run is refused, never
        // paired with a sort of another window width)
        require(shz.c == pk->c_z && shh.c == pk->c_h && (int)shz.sets == pk->s_z && (int)shh.sets == pk->s_h, ZKHIP_ERR_BAD_ARG,
                "the key's tables were built for a window width this build cannot sort: reload the key");
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());   // sums per MSM (2: the tables carry the window multiples)
        sl.ws1.ensure((size_t)4 * Wmax * sizeof(Xyzz<Fq>));   // 4 G1 MSMs + 1 G2 MSM
        sl.ws2.ensure((size_t)Wmax * sizeof(Xyzz<Fq2>));
        Xyzz<Fq>* ws1 = ptr<Xyzz<Fq>>(sl.ws1);
        host_sums(sl, Wmax);
        Xyzz<Fq>* hs1 = (Xyzz<Fq>*)sl.h_ws;                    // the host mirrors of ws1 / ws2: every lane copies its own sums out
        Xyzz<Fq2>* hs2 = (Xyzz<Fq2>*)((uint8_t*)sl.h_ws + (size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        const int lone_sched = (lone && !ctx->serial) ? ctx->lone_sched : 0;
        // An accumulation kernel is sized to fill the machine, saturates the integer multiplier and nothing preempts it:
        // whatever arrives beside it waits for a place or crawls (kernel traces, profiles/r2_single_proof_traces.md: a
        // 0.14 ms mat-vec took 4 ms, a 0.2 ms transform pass 3 ms), h arrives late and the H MSM trails alone behind
        // everything.  So the G2 lane (the longest chain of a proof) starts at once, the witness map runs beside it, and
        // the G1 lanes over z wait for h (`z_gate`; 2 = the G2 lane waits as well).
        const int gate = z_gate(ctx);
        const MsmSort& sort_b = (pk->thin_mask & 8) ? sl.sorts[2] : sl.sorts[0];   // the list b2_ext pairs with (zkhip_pk::thin_mask)
        const bool inf_b2 = (pk->thin_mask & 8) ? pk->inf_many_thin[3] : pk->inf_many[3];
        if (pk->z_n) {
            msm_prepare(ctx, st, sl.sorts[0], (const u32*)d_scalars + pk->z_lo * 8, shz, pk->z_n);
            if (pk->thin_mask) msm_prepare(ctx, st, sl.sorts[2], (const u32*)d_scalars + pk->z_lo * 8, shz, pk->z_n, ptr<u32>(pk->thin_keep), &sl.sorts[0]);
            if (gate < 2) {
                sl.lanes[3].share_cu = (lone_sched & 1) != 0;
                sl.lanes[3].lone_launch = lone;
                msm_run<Fq2>(ctx, sl.lanes[3], sort_b, pk->b2_ext.p, with_inf(shz, inf_b2), ptr<Xyzz<Fq2>>(sl.ws2), sl.acc_b[4], sl.acc_e[4], nullptr, hs2);   // longest first
            }
        }

        // ---- K1-K4 and the h-sort, on the NTT stream: the main stream is free for the next proof's staging and z-sort
        Stream wn = ctx->serial ? st : slot_ntt_stream(ctx, sl);
        stream_wait_event(wn, sl.ev[0]);
        // (lone_sched bit 4: a lone proof's witness map waits for the sort of the assignment — the sort's kernels have the machine to
        // themselves and the G2 accumulation starts that much sooner; the witness map then runs beside it)
        if ((lone_sched & 4) && pk->z_n) stream_wait_event(wn, sl.sorts[0].ready);
        event_record(sl.ev[1], wn);
        ctx->ws = wn;
        witness_map(ctx, cs, pl, bound, half);
        ctx->ws = ctx->stream;
        if (half >= 0) event_record(sl.half_ready, wn);
    }
    // ... and the rest: (the product of the two halves,) the G1 lanes over z, the h sort and the H MSM, the copies out
    static void enqueue_tail(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const zkhip_r1cs* cs) {
        const u64 N = pk->N;
        const bool bound = pk->bound_uid != 0 && pk->bound_uid == cs->uid, lone = sl.lone;
        NttPlan<C>* pl = get_plan<C>(ctx, pk->logN);
        ctx->cur = &sl;
        Stream st = ctx->stream;
        Stream wn = ctx->serial ? st : slot_ntt_stream(ctx, sl);
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        Xyzz<Fq>* ws1 = ptr<Xyzz<Fq>>(sl.ws1);
        Xyzz<Fq>* hs1 = (Xyzz<Fq>*)sl.h_ws;
        Xyzz<Fq2>* hs2 = (Xyzz<Fq2>*)((uint8_t*)sl.h_ws + (size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        const int lone_sched = (lone && !ctx->serial) ? ctx->lone_sched : 0;
        const int gate = z_gate(ctx);
        const MsmSort& sort_b = (pk->thin_mask & 8) ? sl.sorts[2] : sl.sorts[0];
        const bool inf_b2 = (pk->thin_mask & 8) ? pk->inf_many_thin[3] : pk->inf_many[3];
        // where the H MSM's scalars are: the first vector — except for a split proof, whose product goes to the THIRD (c's, idle over a
        // bound key): its own half stays as it is for the partner that may still be copying it
        const u32* h_scalars = ptr<u32>(sl.va) + (sl.half >= 0 ? 2 * N * 8 : 0);
        if (sl.half >= 0) {
            // both vectors are on the coset now (the other one arrived on this stream): U_j = a_j b_j / Z(g), canonical integers
            Fr* a = ptr<Fr>(sl.va);
            ZK_LAUNCH((k_quotient<typename Fr::Params>), dim3(blocks_for(N, 256)), dim3(256), 0, wn, a, a + N, fe_from_mont(pl->zinv), a + 2 * N, N);
        }
        event_record(sl.ev[2], wn);

        // (lone_sched bit 2: the h sort goes out BEFORE the G1 lanes over z, which then wait for it — it runs beside the G2 lane alone)
        const bool h_sort_first = (lone_sched & 2) != 0 && pk->h_n && gate;
        if (h_sort_first) msm_prepare(ctx, wn, sl.sorts[1], h_scalars + pk->h_lo * 8, shh, pk->h_n);
        if (pk->z_n) {
            const Event h_ready = !gate ? nullptr : h_sort_first ? sl.sorts[1].ready : sl.ev[2];
            if (gate >= 2) {
                sl.lanes[3].lone_launch = lone;
                msm_run<Fq2>(ctx, sl.lanes[3], sort_b, pk->b2_ext.p, with_inf(shz, inf_b2), ptr<Xyzz<Fq2>>(sl.ws2), sl.acc_b[4], sl.acc_e[4], h_ready, hs2);
            }
            // A G2 accumulation at one wave per SIMD (BLS12-381: the wave takes the SIMD's whole register file) shares no SIMD with a
            // G1 wave: G1 workgroups that arrive while some of its workgroups are still waiting for a place take the places, and the
            // G2 lane — the longest chain of such a proof — finishes that much later.  The witness map used to be the head start; a
            // bound key's is a third shorter, and the lone Poseidon proof went from 6.9 to 8.1 ms
            // (profiles/r5k_bound_key_single_proof_latency_ab.txt).  So a LONE proof holds its G1 lanes until the G2 accumulation is
            // through; a batch has other proofs' kernels to fill the machine and is left alone.
            Event g1_after = h_ready;
            constexpr bool g2_owns_simds = MsmTuning<typename Unsat<Fq2>::type>::ACCUM_WPE == 1;
            if (g2_owns_simds && lone && !ctx->serial && (ctx->g2_head_start == 2 || (ctx->g2_head_start == 1 && bound))) {
                if (h_ready) stream_wait_event(st, h_ready);
                stream_wait_event(st, sl.acc_e[4]);
                event_record(sl.g1_go, st);
                g1_after = sl.g1_go;
            }
            if (lone) sl.lanes[0].lone_launch = sl.lanes[1].lone_launch = sl.lanes[2].lone_launch = true;
            run_z_g1(ctx, sl, pk, shz, ws1, Wmax, g1_after, bound, hs1);
        } else {
            empty_msm(ctx, sl, ws1, 3 * Wmax, ptr<Xyzz<Fq2>>(sl.ws2), Wmax, 0, 4);
        }

        // ---- H = MSM(h_query, h) in sigma order (the zero-padded tail pairs with infinity bases)
        if (pk->h_n) {
            if (!h_sort_first) msm_prepare(ctx, wn, sl.sorts[1], h_scalars + pk->h_lo * 8, shh, pk->h_n);
            // (a bound key: U in natural order against H' — the same MSM machinery, other bases)
            if (lone) sl.lanes[4].lone_launch = true;
            msm_run<Fq>(ctx, sl.lanes[4], sl.sorts[1], bound ? pk->h_bound.p : pk->h_sigma.p, with_inf(shh, bound ? pk->inf_many_bound[1] : pk->inf_many[4]),
                        ws1 + 3 * Wmax, sl.acc_b[3], sl.acc_e[3], nullptr, hs1 + 3 * Wmax);
        } else {
            empty_msm(ctx, sl, ws1 + 3 * Wmax, Wmax, nullptr, 0, 4, 5);
        }

        copy_out(ctx, sl, Wmax);
    }

    // ---- A, B1, L: the three G1 MSMs over the sorted assignment (window sums to ws1 + {0, 1, 2} * Wmax)
    // (`bound`: L' in l's place, on the common list whatever l's family is — its public entries are finite)
    static void run_z_g1(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const MsmShape& shz, Xyzz<Fq>* ws1, int Wmax, Event h_ready, bool bound = false,
                         Xyzz<Fq>* hs1 = nullptr) {
        const void* tab[3] = {pk->a_ext.p, pk->b1_ext.p, bound ? pk->l_bound.p : pk->l_ext.p};
        auto thin = [&](int k) { return (bound && k == 2) ? 0u : (pk->thin_mask >> k) & 1u; };
        auto inf_many = [&](int k) { return (bound && k == 2) ? pk->inf_many_bound[0] : thin(k) ? pk->inf_many_thin[k] : pk->inf_many[k]; };
        if (!ctx->fuse_z) {
            for (int k = 0; k < 3; ++k)
                msm_run<Fq>(ctx, sl.lanes[k], thin(k) ? sl.sorts[2] : sl.sorts[0], tab[k], with_inf(shz, inf_many(k)), ws1 + k * Wmax, sl.acc_b[k], sl.acc_e[k], h_ready,
                            hs1 ? hs1 + k * Wmax : nullptr);
            return;
        }
        // One launch per sorted list: the tables on the common list together, the tables on the thinned list together (up to three
        // tables on one, none on the other).  The first table of a group lends its lane; the others ride on its launches.
        for (u32 which = 0; which < 2; ++which) {
            int member[3], nm = 0;
            for (int k = 0; k < 3; ++k)
                if (thin(k) == which) member[nm++] = k;
            if (!nm) continue;
            const int lead = member[0];
            const void* tabs[3];
            bool many = false;
            for (int q = 0; q < nm; ++q) { tabs[q] = tab[member[q]]; many = many || inf_many(member[q]); }
            // destination slots: lead, then every `step` slots (any subset of {0, 1, 2} is an arithmetic progression)
            const int step = nm > 1 ? member[1] - member[0] : 1;
            msm_run_tables<Fq>(ctx, sl.lanes[lead], which ? sl.sorts[2] : sl.sorts[0], tabs, nm, with_inf(shz, many), ws1 + lead * Wmax, (u32)(step * Wmax),
                               sl.acc_b[lead], sl.acc_e[lead], h_ready, hs1 ? hs1 + lead * Wmax : nullptr);
            Stream s0 = ctx->serial ? ctx->stream : lane_stream(sl.lanes[lead]);
            for (int q = 1; q < nm; ++q) {
                event_record(sl.acc_b[member[q]], s0);
                event_record(sl.acc_e[member[q]], s0);
                event_record(sl.lanes[member[q]].done, s0);
            }
        }
    }

    // the slot's pinned host copy of the window sums (4 G1 MSMs, the G2 MSM, the verdict word of the canonical check)
    static void host_sums(ProofSlot& sl, int Wmax) {
        const size_t b1 = (size_t)4 * Wmax * sizeof(Xyzz<Fq>), b2 = (size_t)Wmax * sizeof(Xyzz<Fq2>);
        if (sl.h_ws_cap < b1 + b2 + 4) {
            host_free_pinned(sl.h_ws);
            sl.h_ws = nullptr; sl.h_ws_cap = 0;
            sl.h_ws = host_alloc_pinned(b1 + b2 + 4);
            sl.h_ws_cap = b1 + b2 + 4;
        }
    }
    // ---- the end of a proof's device work: every lane has copied its window sums out behind its fold (msm_run_tables) and recorded
    // `done`; a stream of its own — the main stream is free for the next proof — waits for all of them, fetches the verdict of the
    // canonical check and records "all done"
    static void copy_out(zkhip_ctx* ctx, ProofSlot& sl, int Wmax) {
        Stream st = ctx->stream;
        Stream so = ctx->serial ? st : ctx_out_stream(ctx);
        for (int k = 0; k < ZK_NLANES; ++k) stream_wait_event(so, sl.lanes[k].done);
        const size_t b1 = (size_t)4 * Wmax * sizeof(Xyzz<Fq>), b2 = (size_t)Wmax * sizeof(Xyzz<Fq2>);
        sl.zflag.ensure(4);
        dev_d2h_pinned((uint8_t*)sl.h_ws + b1 + b2, sl.zflag.p, 4, so);   // written on the main stream before ev[0], which every lane waits for
        event_record(sl.ev[3], so);
        sl.busy = true;
    }

    // a rank whose range is empty (more ranks than points): all-infinity window sums, events recorded so that the
    // bookkeeping of the slot stays uniform
    static void empty_msm(zkhip_ctx* ctx, ProofSlot& sl, Xyzz<Fq>* ws1, size_t n1, Xyzz<Fq2>* ws2, size_t n2, int lane_from, int lane_to) {
        Stream st = ctx->stream;
        if (n1) {
            dev_memset(ws1, 0, n1 * sizeof(Xyzz<Fq>), st);
            memset((uint8_t*)sl.h_ws + ((const uint8_t*)ws1 - (const uint8_t*)sl.ws1.p), 0, n1 * sizeof(Xyzz<Fq>));     // (the host mirror: nothing is in flight for these slots)
        }
        if (n2) {
            dev_memset(ws2, 0, n2 * sizeof(Xyzz<Fq2>), st);
            memset((uint8_t*)sl.h_ws + (size_t)4 * n2 * sizeof(Xyzz<Fq>) + ((const uint8_t*)ws2 - (const uint8_t*)sl.ws2.p), 0, n2 * sizeof(Xyzz<Fq2>));   // (n2 = Wmax: the G2 sums follow the 4 x Wmax G1 sums)
        }
        for (int k = lane_from; k < lane_to; ++k) {
            const int e = k == 3 ? 4 : k == 4 ? 3 : k;   // lane 3 (B2) times with event pair 4, lane 4 (H) with pair 3
            event_record(sl.acc_b[e], st);
            event_record(sl.acc_e[e], st);
            event_record(sl.lanes[k].done, st);
        }
    }

    // the five MSM results of a proof (or of one rank's share of it)
    struct Sums {
        Xyzz<Fq> a, b1, l, h;
        Xyzz<Fq2> b2;
    };
    // ---- wait for the slot's proof; Horner over its window sums
    static Sums collect(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk) {
        require(sl.busy, ZKHIP_ERR_DEVICE, "internal: no proof in flight in this slot");
        event_sync(sl.ev[3]);
        sl.busy = false;
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        const Xyzz<Fq>* h_ws1 = (const Xyzz<Fq>*)sl.h_ws;
        const Xyzz<Fq2>* h_ws2 = (const Xyzz<Fq2>*)((const uint8_t*)sl.h_ws + (size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        u32 zflag;
        memcpy(&zflag, (const uint8_t*)(h_ws2 + Wmax), 4);
        require_canonical(zflag);
        // five independent Horner chains (W x c doublings each): one host thread per MSM, the G2 chain on this one
        Sums g;
        HostThreads th;
        th.run([&] { g.a = msm_combine(&h_ws1[0 * Wmax], shz); });
        th.run([&] { g.b1 = msm_combine(&h_ws1[1 * Wmax], shz); });
        th.run([&] { g.l = msm_combine(&h_ws1[2 * Wmax], shz); });
        th.run([&] { g.h = msm_combine(&h_ws1[3 * Wmax], shh); });
        g.b2 = msm_combine(h_ws2, shz);
        th.join();
        return g;
    }
    // ---- K9: C = s*A + r*B1 - rs*delta_1 + L + H   (App. A.3; alpha/beta/delta terms already inside A, B1, B2)
    static void assemble(const zkhip_pk* pk, const Sums& g, const uint8_t* r, const uint8_t* s_, uint8_t* out) {
        const Fr rr = fe_from_bytes_canon<Fr>(r), ss = fe_from_bytes_canon<Fr>(s_);
        // three independent 254-bit scalar multiplications
        Xyzz<Fq> sA, rB1, rsD;
        {
            HostThreads th;
            th.run([&] { sA = xyzz_mul_limbs(g.a, ss.v, Fr::N); });
            th.run([&] { rB1 = xyzz_mul_limbs(g.b1, rr.v, Fr::N); });
            rsD = rs_delta(pk, rr, ss);
        }
        assemble_tail(g, sA, rB1, rsD, out);
    }
    static void assemble_tail(const Sums& g, const Xyzz<Fq>& sA, const Xyzz<Fq>& rB1, const Xyzz<Fq>& rsD, uint8_t* out) {
        const Xyzz<Fq>&gA = g.a, &gL = g.l, &gH = g.h;
        const Xyzz<Fq2>& gB2 = g.b2;
        Xyzz<Fq> gC = xyzz_add(sA, rB1);
        gC = xyzz_add(gC, xyzz_neg(rsD));
        gC = xyzz_add(gC, gL);
        gC = xyzz_add(gC, gH);
        Aff<Fq> pa, pc;
        Aff<Fq2> pb;
        {
            HostThreads th;               // three inversions: one thread each
            th.run([&] { pa = xyzz_to_affine(gA); });
            th.run([&] { pc = xyzz_to_affine(gC); });
            pb = xyzz_to_affine(gB2);
        }
        memset(out, 0, 8 * FQB + 3);
        if (!gA.is_inf()) { write_fe(pa.x, out); write_fe(pa.y, out + FQB); }
        if (!gB2.is_inf()) {
            write_fe(pb.x.c0, out + 2 * FQB); write_fe(pb.x.c1, out + 3 * FQB);
            write_fe(pb.y.c0, out + 4 * FQB); write_fe(pb.y.c1, out + 5 * FQB);
        }
        if (!gC.is_inf()) { write_fe(pc.x, out + 6 * FQB); write_fe(pc.y, out + 7 * FQB); }
        out[8 * FQB] = gA.is_inf(); out[8 * FQB + 1] = gB2.is_inf(); out[8 * FQB + 2] = gC.is_inf();
    }
    static void fill_timings(ProofSlot& sl, zkhip_timings* tm, std::chrono::steady_clock::time_point t_fin) {
        const auto t_end = std::chrono::steady_clock::now();
        if (tm) {
            memset(tm, 0, sizeof(*tm));
            // the five MSMs run on their own streams, concurrently with each other and with the NTT pipeline:
            // msm_z = staging done -> last of A/B1/L/B2 finished; msm_h = h ready -> H finished (overlapping intervals)
            for (int k = 0; k < 4; ++k) tm->msm_z_ms = std::max(tm->msm_z_ms, event_elapsed_ms(sl.ev[0], sl.lanes[k].done));
            tm->ntt_ms = event_elapsed_ms(sl.ev[1], sl.ev[2]);   // matvec + 7 transforms + quotient
            tm->kernel_ntt_ms = event_elapsed_ms(sl.ntt_b, sl.ntt_e);   // the transform passes and the quotient kernel only
            tm->msm_h_ms = event_elapsed_ms(sl.ev[2], sl.lanes[4].done);
            tm->finish_ms = std::chrono::duration<float, std::milli>(t_end - t_fin).count();
            tm->total_ms = std::chrono::duration<float, std::milli>(t_end - sl.t_start).count();
            for (int k = 0; k < 4; ++k) tm->kernel_msm_accum_g1_ms += event_elapsed_ms(sl.acc_b[k], sl.acc_e[k]);
            tm->kernel_msm_accum_g2_ms = event_elapsed_ms(sl.acc_b[4], sl.acc_e[4]);
        }
    }
    // ---- the host's share of a proof, taken in the order the lanes deliver: r s delta_1 needs nothing of the device, s A and r B1 wait
    // for the lanes over z only — three 254-bit scalar multiplications (0.2 ms) that used to start after the LAST lane and now run
    // beside the H MSM's tail; what is left behind the last event is two group additions and three conversions to affine form
    static void finish(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, uint8_t* out, zkhip_timings* tm) {
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_g16_partial + zkhip_combine_g16");
        require(sl.busy, ZKHIP_ERR_DEVICE, "internal: no proof in flight in this slot");
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        const Xyzz<Fq>* h_ws1 = (const Xyzz<Fq>*)sl.h_ws;
        const Xyzz<Fq2>* h_ws2 = (const Xyzz<Fq2>*)((const uint8_t*)sl.h_ws + (size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        const Fr rr = fe_from_bytes_canon<Fr>(sl.r), ss = fe_from_bytes_canon<Fr>(sl.s);
        Sums g;
        Xyzz<Fq> sA, rB1, rsD;
        std::chrono::steady_clock::time_point t_fin;
        try {
            HostThreads th;
            th.run([&] { rsD = rs_delta(pk, rr, ss); });
            event_sync(sl.lanes[3].done);
            g.b2 = msm_combine(h_ws2, shz);
            for (int k = 0; k < 3; ++k) event_sync(sl.lanes[k].done);
            g.a = msm_combine(&h_ws1[0 * Wmax], shz);
            g.b1 = msm_combine(&h_ws1[1 * Wmax], shz);
            th.run([&] { sA = xyzz_mul_limbs(g.a, ss.v, Fr::N); });
            th.run([&] { rB1 = xyzz_mul_limbs(g.b1, rr.v, Fr::N); });
            g.l = msm_combine(&h_ws1[2 * Wmax], shz);
            event_sync(sl.lanes[4].done);
            t_fin = std::chrono::steady_clock::now();
            g.h = msm_combine(&h_ws1[3 * Wmax], shh);
            event_sync(sl.ev[3]);
            th.join();
        } catch (...) {
            sl.busy = false;
            throw;
        }
        sl.busy = false;
        u32 zflag;
        memcpy(&zflag, (const uint8_t*)(h_ws2 + Wmax), 4);
        require_canonical(zflag);
        assemble_tail(g, sA, rB1, rsD, out);
        fill_timings(sl, tm, t_fin);
    }
    static Xyzz<Fq> rs_delta(const zkhip_pk* pk, const Fr& rr, const Fr& ss) {
        const Fr rs = fe_from_mont(fe_mul(fe_to_mont(rr), fe_to_mont(ss)));
        uint8_t dec[2 * FQB];
        decode_point<FQB, 2>(pk->delta_g1_canon.data(), dec);
        Aff<Fq> d1;
        memcpy(&d1, dec, sizeof(d1));
        d1 = PkLoader<C>::to_mont_point(d1);
        return xyzz_mul_limbs(Xyzz<Fq>::from_affine(d1), rs.v, Fr::N);
    }
    // The canonical representative of a group element as an XYZZ record: affine coordinates with ZZ = ZZZ = 1 (Montgomery
    // one), infinity all-zero.  The projective coordinates an MSM leaves depend on the order in which the bucket sort's
    // atomics placed the points, i.e. they vary from run to run and from machine to machine while the group element does
    // not; records that cross a process or machine boundary (zkhip_prove_*_partial) are therefore normalised first, so that
    // equal partial sums are equal bytes.  Five host inversions per proof share.
    template <class F>
    static Xyzz<F> canonical(const Xyzz<F>& p) {
        if (p.is_inf()) return Xyzz<F>::inf();
        const Aff<F> a = xyzz_to_affine(p);
        return {a.x, a.y, F::one(), F::one()};
    }
    static void canonicalise(Sums& g) {
        HostThreads th;
        th.run([&] { g.a = canonical(g.a); });
        th.run([&] { g.b1 = canonical(g.b1); });
        th.run([&] { g.l = canonical(g.l); });
        th.run([&] { g.h = canonical(g.h); });
        g.b2 = canonical(g.b2);
        th.join();
    }
    // one rank's share of a proof: the five partial sums as canonical XYZZ records (saturated Montgomery limbs)
    static void prove_partial(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* z_dev, const uint8_t* r,
                              const uint8_t* s_, uint8_t* partial_out, zkhip_timings* tm) {
        enqueue(ctx, ctx->slots[0], pk, cs, z_host, z_dev, r, s_);
        Sums g = collect(ctx, ctx->slots[0], pk);
        const auto t_fin = std::chrono::steady_clock::now();
        canonicalise(g);
        memcpy(partial_out, &g, sizeof(g));
        fill_timings(ctx->slots[0], tm, t_fin);
    }
    // ---- a member's share of a proof whose witness map is SPLIT between the members (a bound key: the proof needs a and b on the
    // coset; the members of even rank transform a, those of odd rank b, and partners exchange — north_star's "NTT domain shard"):
    //   split_begin     the head with this member's half;
    //   split_fetch*    the partner's half into this member's vectors (device to device across xGMI / from host memory);
    //   split_half_out  this member's half to host memory (the multi-process exchange);
    //   split_end_*     the tail, and the member's result as prove_partial / prove_device_sums leave it.
    static void split_begin(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* z_dev, const uint8_t* r,
                            const uint8_t* s_, int half) {
        require(pk->bound_uid != 0 && pk->bound_uid == cs->uid && pk->scheme == 0 && (half == 0 || half == 1), ZKHIP_ERR_BAD_ARG,
                "only a Groth16 proof over a key bound to its constraint system splits its witness map (half = 0: a, 1: b)");
        enqueue_head(ctx, ctx->slots[0], pk, cs, z_host, z_dev, r, s_, false, half);
    }
    static const void* split_half_ptr(zkhip_ctx* ctx, const zkhip_pk* pk, int half) { return ptr<Fr>(ctx->slots[0].va) + (u64)half * pk->N; }
    static void split_fetch(zkhip_ctx* ctx, const zkhip_pk* pk, const void* src, int src_device, Event src_ready) {
        ProofSlot& sl = ctx->slots[0];
        require(sl.half >= 0, ZKHIP_ERR_BAD_ARG, "internal: no split proof in flight");
        Stream wn = ctx->serial ? ctx->stream : ctx_ntt_stream(ctx);
        if (src_ready) event_sync(src_ready);           // (the partner's stream, possibly on another device: waited for on the host)
        dev_copy_between(ptr<Fr>(sl.va) + (u64)(1 - sl.half) * pk->N, ctx->device, src, src_device, pk->N * sizeof(Fr), wn);
    }
    static void split_fetch_host(zkhip_ctx* ctx, const zkhip_pk* pk, const uint8_t* other_half) {
        ProofSlot& sl = ctx->slots[0];
        require(sl.half >= 0, ZKHIP_ERR_BAD_ARG, "internal: no split proof in flight");
        Stream wn = ctx->serial ? ctx->stream : ctx_ntt_stream(ctx);
        dev_h2d(ptr<Fr>(sl.va) + (u64)(1 - sl.half) * pk->N, other_half, pk->N * sizeof(Fr), wn);
    }
    static void split_half_out(zkhip_ctx* ctx, const zkhip_pk* pk, uint8_t* out) {
        ProofSlot& sl = ctx->slots[0];
        require(sl.half >= 0, ZKHIP_ERR_BAD_ARG, "internal: no split proof in flight");
        Stream wn = ctx->serial ? ctx->stream : ctx_ntt_stream(ctx);
        dev_d2h(out, ptr<Fr>(sl.va) + (u64)sl.half * pk->N, pk->N * sizeof(Fr), wn);
        stream_sync(wn);
    }
    static void split_end_partial(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, uint8_t* partial_out, zkhip_timings* tm) {
        ProofSlot& sl = ctx->slots[0];
        require(sl.half >= 0, ZKHIP_ERR_BAD_ARG, "internal: no split proof in flight");
        enqueue_tail(ctx, sl, pk, cs);
        Sums g = collect(ctx, sl, pk);
        const auto t_fin = std::chrono::steady_clock::now();
        canonicalise(g);
        memcpy(partial_out, &g, sizeof(g));
        fill_timings(sl, tm, t_fin);
    }
    static void split_end_device_sums(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const void** d_ws1, size_t* b1, const void** d_ws2, size_t* b2,
                                      zkhip_timings* tm) {
        ProofSlot& sl = ctx->slots[0];
        require(sl.half >= 0, ZKHIP_ERR_BAD_ARG, "internal: no split proof in flight");
        enqueue_tail(ctx, sl, pk, cs);
        wait_device_sums(ctx, sl, pk, d_ws1, b1, d_ws2, b2);
        fill_timings(sl, tm, std::chrono::steady_clock::now());
    }
    // one rank's share of a proof, left ON THE DEVICE: the raw bucket-set sums of its five MSMs (ws1: 4 G1 MSMs x Wmax XYZZ
    // sums, ws2: the G2 MSM), for an exchange that never touches host memory (zkhip_multi_use_rccl: RCCL all-gather over
    // xGMI).  Valid until the slot's next proof.
    static void prove_device_sums(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const uint8_t* r,
                                  const uint8_t* s_, const void** d_ws1, size_t* b1, const void** d_ws2, size_t* b2, zkhip_timings* tm) {
        ProofSlot& sl = ctx->slots[0];
        enqueue(ctx, sl, pk, cs, z_host, nullptr, r, s_);
        wait_device_sums(ctx, sl, pk, d_ws1, b1, d_ws2, b2);
        fill_timings(sl, tm, std::chrono::steady_clock::now());
    }
    static void wait_device_sums(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const void** d_ws1, size_t* b1, const void** d_ws2, size_t* b2) {
        require(sl.busy, ZKHIP_ERR_DEVICE, "internal: no proof in flight in this slot");
        event_sync(sl.ev[3]);
        sl.busy = false;
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        *b1 = (size_t)4 * Wmax * sizeof(Xyzz<Fq>);
        *b2 = (size_t)Wmax * sizeof(Xyzz<Fq2>);
        u32 zflag;
        memcpy(&zflag, (const uint8_t*)sl.h_ws + *b1 + *b2, 4);
        require_canonical(zflag);
        *d_ws1 = sl.ws1.p;
        *d_ws2 = sl.ws2.p;
    }
    // the same sums, gathered into host memory (one rank's ws1 | ws2), as a partial record for `combine` (not canonical:
    // these never leave the process)
    static void record_from_sums(zkhip_ctx* ctx, const zkhip_pk* pk, const uint8_t* ws1, const uint8_t* ws2, uint8_t* record_out) {
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        std::vector<Xyzz<Fq>> a1((size_t)4 * Wmax);
        std::vector<Xyzz<Fq2>> a2((size_t)Wmax);
        memcpy(a1.data(), ws1, a1.size() * sizeof(Xyzz<Fq>));
        memcpy(a2.data(), ws2, a2.size() * sizeof(Xyzz<Fq2>));
        Sums g;
        g.a = msm_combine(&a1[0 * Wmax], shz);
        g.b1 = msm_combine(&a1[1 * Wmax], shz);
        g.l = msm_combine(&a1[2 * Wmax], shz);
        g.h = msm_combine(&a1[3 * Wmax], shh);
        g.b2 = msm_combine(a2.data(), shz);
        memcpy(record_out, &g, sizeof(g));
    }
    // sum of the ranks' partial results, then the assembly
    static void combine(const zkhip_pk* pk, u32 count, const uint8_t* partials, const uint8_t* r, const uint8_t* s_, uint8_t* out) {
        Sums t;
        t.a = t.b1 = t.l = t.h = Xyzz<Fq>::inf();
        t.b2 = Xyzz<Fq2>::inf();
        for (u32 i = 0; i < count; ++i) {
            Sums g;
            memcpy(&g, partials + (size_t)i * sizeof(Sums), sizeof(g));
            t.a = xyzz_add(t.a, g.a); t.b1 = xyzz_add(t.b1, g.b1); t.l = xyzz_add(t.l, g.l); t.h = xyzz_add(t.h, g.h);
            t.b2 = xyzz_add(t.b2, g.b2);
        }
        Fr rr = fe_from_bytes_canon<Fr>(r), ss = fe_from_bytes_canon<Fr>(s_);
        require(canon_lt_mod(rr) && canon_lt_mod(ss), ZKHIP_ERR_BAD_ARG, "r or s not a canonical field element");
        assemble(pk, t, r, s_, out);
    }
    static constexpr size_t PARTIAL_BYTES = sizeof(Sums);

    // one proof from a host assignment / from an assignment resident in HBM
    static void prove_host(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z, const uint8_t* r, const uint8_t* s_,
                           uint8_t* out, zkhip_timings* tm) {
        enqueue(ctx, ctx->slots[0], pk, cs, z, nullptr, r, s_, true);
        finish(ctx, ctx->slots[0], pk, out, tm);
    }
    static void prove_resident(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, void* d_scalars, const uint8_t* r, const uint8_t* s_,
                               uint8_t* out, zkhip_timings* tm) {
        enqueue(ctx, ctx->slots[0], pk, cs, nullptr, d_scalars, r, s_, true);
        finish(ctx, ctx->slots[0], pk, out, tm);
    }
    // `count` proofs, two in flight: while the GPU works on proof i the host finishes proof i-1 and enqueues proof i+1,
    // so the latency-bound tail of one proof (the H fold) overlaps the MSMs of the next.
    // z_host: count x m x 32 B, or nullptr with z_dev[i] = device pointers of resident assignments.
    static void prove_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, u32 count, const uint8_t* z_host, void* const* z_dev,
                            const uint8_t* rs, uint8_t* proofs_out, zkhip_timings* tm) {
        const size_t proof_bytes = 8 * FQB + 3;
        zkhip_timings acc, one;
        memset(&acc, 0, sizeof(acc));
        const auto t0 = std::chrono::steady_clock::now();
        try {
            const u32 NS = (u32)ctx->nslots;   // proofs in flight
            for (u32 i = 0; i < count + NS - 1; ++i) {
                if (i < count)
                    enqueue(ctx, ctx->slots[i % NS], pk, cs, z_host ? z_host + (size_t)i * pk->m * 32 : nullptr, z_host ? nullptr : z_dev[i],
                            rs + (size_t)i * 64, rs + (size_t)i * 64 + 32);
                if (i >= NS - 1 && i - (NS - 1) < count) {
                    const u32 j = i - (NS - 1);
                    finish(ctx, ctx->slots[j % NS], pk, proofs_out + (size_t)j * proof_bytes, &one);
                    float* a = (float*)&acc; const float* b = (const float*)&one;
                    for (size_t k = 0; k < sizeof(acc) / sizeof(float); ++k) a[k] += b[k];
                }
            }
        } catch (...) {
            for (auto& sl : ctx->slots) sl.busy = false;   // let the queues drain; the context stays usable
            dev_sync_all();
            throw;
        }
        if (tm) {
            *tm = acc;
            tm->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    static void assignment_upload(zkhip_ctx* ctx, zkhip_assignment* a, const uint8_t* z) {
        DBuf flag;
        upload_z(ctx, a->scalars, a->m, z, flag);
        u32 verdict = 0;
        dev_d2h(&verdict, flag.p, 4, ctx->stream);
        stream_sync(ctx->stream);
        require_canonical(verdict);
    }

    // generic MSM primitive (bases in ark encoding)
    template <class F, int NC>
    static void msm_api(zkhip_ctx* ctx, u64 n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out) {
        constexpr int PB = FQB * NC;
        memset(out, 0, PB + 1);
        if (n == 0) { out[PB] = 1; return; }
        require(n < ((u64)1 << 31), ZKHIP_ERR_BAD_ARG, "too many points");
        Stream s = ctx->stream;
        std::vector<uint8_t> host(n * PB);
        for (u64 i = 0; i < n; ++i) decode_point<FQB, NC>(bases + i * PB, &host[i * PB]);
        DBuf d_bases, d_ws;
        d_bases.ensure(host.size());
        dev_h2d(d_bases.p, host.data(), host.size(), s);
        ZK_LAUNCH((k_to_mont<Fq>), dim3(blocks_for(n * NC, 256)), dim3(256), 0, s, ptr<Fq>(d_bases), ptr<Fq>(d_bases), n * NC);
        ctx->cur->scalars.ensure(n * 32);
        dev_h2d(ctx->cur->scalars.p, scalars, n * 32, s);
        // ad-hoc bases: no table of window multiples (building one costs ~15x the MSM itself), one bucket set per window
        const MsmShape sh = msm_shape(ctx, n, Fr::Params::BITS, false);
        d_ws.ensure((size_t)sh.nsums() * sizeof(Xyzz<F>));
        // the packed bases are written on the main stream BEFORE the sort records `ready` there: the lane stream waits on
        // that event only, so everything the accumulation reads must precede it on the main stream
        DBuf d_packed;
        d_packed.ensure(n * packed_point_bytes<F>());
#ifdef ZK_TEST_MSM_RACE   // round 2's ordering, kept ONLY to show that tests/test_stream_jitter.py catches it (never built into libzkhip.so)
        msm_prepare(ctx, s, ctx->cur->sorts[0], ptr<u32>(ctx->cur->scalars), sh, 0);
        points_to_packed<F>(ctx, ptr<Aff<F>>(d_bases), d_packed.p, n);
#else
        points_to_packed<F>(ctx, ptr<Aff<F>>(d_bases), d_packed.p, n);
        msm_prepare(ctx, s, ctx->cur->sorts[0], ptr<u32>(ctx->cur->scalars), sh, 0);
#endif
        msm_run<F>(ctx, ctx->cur->lanes[0], ctx->cur->sorts[0], d_packed.p, sh, ptr<Xyzz<F>>(d_ws), nullptr, nullptr);
        stream_wait_event(s, ctx->cur->lanes[0].done);
        std::vector<Xyzz<F>> ws(sh.nsums());
        dev_d2h(ws.data(), d_ws.p, ws.size() * sizeof(Xyzz<F>), s);
        stream_sync(s);
        Xyzz<F> res = msm_combine(ws.data(), sh);
        if (res.is_inf()) { out[PB] = 1; return; }
        Aff<F> a = xyzz_to_affine(res);
        uint8_t tmp[sizeof(Aff<F>)];
        Aff<F> canon = from_mont_point(a);
        memcpy(tmp, &canon, sizeof(canon));
        memcpy(out, tmp, PB);
    }
    static Aff<Fq> from_mont_point(const Aff<Fq>& p) { return {fe_from_mont(p.x), fe_from_mont(p.y)}; }
    static Aff<Fq2> from_mont_point(const Aff<Fq2>& p) { return {fe_from_mont(p.x), fe_from_mont(p.y)}; }

    static void ntt_api(zkhip_ctx* ctx, u32 log_n, int dir, uint8_t* data) {
        NttPlan<C>* pl = get_plan<C>(ctx, (int)log_n);
        Stream s = ctx->stream;
        ctx->ws = s;
        const u64 N = pl->N;
        const unsigned T = 256, B = blocks_for(N, T);
        ctx->cur->va.ensure(N * sizeof(Fr));
        ctx->cur->vb.ensure(N * sizeof(Fr));
        ctx->cur->vc.ensure(N * sizeof(Fr));
        Fr *a = ptr<Fr>(ctx->cur->va), *b = ptr<Fr>(ctx->cur->vb), *t = ptr<Fr>(ctx->cur->vc);
        dev_h2d(a, data, N * 32, s);
        ZK_LAUNCH((k_mul_const<Fr>), dim3(B), dim3(T), 0, s, a, a, N, pl->k_to_rp);   // canonical -> R'-form
        const bool inverse = dir == 1 || dir == 3;
        if (dir == 2) {   // coset_fft: x_i * g^i first (a saturated-Montgomery factor keeps the R'-form)
            ZK_LAUNCH((k_pow_table<Fr>), dim3(B), dim3(T), 0, s, t, pl->g, Fr::one(), N, 0u, 0u, 0);
            ZK_LAUNCH((k_mul_table<Fr>), dim3(B), dim3(T), 0, s, a, t, a, N);
        }
        ntt_kind_a<C>(ctx, pl, a, inverse, nullptr, 1, 0, 1);
        ZK_LAUNCH((k_sigma_permute<Fr>), dim3(B), dim3(T), 0, s, a, b, N, pl->N1, pl->N2, 1, pl->N3);
        if (inverse) {    // * 1/N (and g^-i for coset_ifft)
            ZK_LAUNCH((k_pow_table<Fr>), dim3(B), dim3(T), 0, s, t, dir == 3 ? pl->g_inv : Fr::one(), pl->n_inv, N, 0u, 0u, 0);
            ZK_LAUNCH((k_mul_table<Fr>), dim3(B), dim3(T), 0, s, b, t, b, N);
        }
        ZK_LAUNCH((k_from_rp<Fr>), dim3(B), dim3(T), 0, s, b, b, N);
        dev_d2h(data, b, N * 32, s);
        stream_sync(s);
        // A stand-alone transform of a very large domain leaves 8 N x 32 bytes behind (five factor tables of its plan, three work
        // vectors: 69 GiB at 2^28) that nothing else will use at that size: given back at once.  A prover at such a domain keeps its
        // plan (it is rebuilt in about a second if a transform of this kind evicted it).
        if (log_n > 24) {
            ctx->cur->va.release(); ctx->cur->vb.release(); ctx->cur->vc.release();
            for (size_t k = 0; k < ctx->plans.size(); ++k)
                if (ctx->plans[k].get() == pl) { ctx->plans.erase(ctx->plans.begin() + (long)k); break; }
        }
    }

    static void witness_map_api(zkhip_ctx* ctx, const zkhip_r1cs* cs, const uint8_t* z, uint8_t* h_out) {
        ctx->ws = ctx->stream;
        NttPlan<C>* pl = get_plan<C>(ctx, cs->logN);
        const u64 m = cs->l + cs->w;
        uint8_t zero[32] = {0};
        upload_z(ctx, ctx->cur->scalars, m, z, ctx->cur->zflag);
        stage_scalars(ctx, pl, ctx->cur->scalars.p, m, zero, zero);
        witness_map(ctx, cs, pl);
        ctx->cur->vb.ensure(pl->N * sizeof(Fr));
        ZK_LAUNCH((k_sigma_permute<Fr>), dim3(blocks_for(pl->N, 256)), dim3(256), 0, ctx->stream, ptr<Fr>(ctx->cur->va), ptr<Fr>(ctx->cur->vb), pl->N,
                  pl->N1, pl->N2, 1, pl->N3);
        dev_d2h(h_out, ctx->cur->vb.p, pl->N * 32, ctx->stream);
        stream_sync(ctx->stream);
    }

    template <class F>
    static void field_op_api(zkhip_ctx* ctx, int op, u64 count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
        Stream s = ctx->stream;
        const size_t bytes = count * sizeof(F);
        ctx->cur->va.ensure(bytes); ctx->cur->vb.ensure(bytes);
        dev_h2d(ctx->cur->va.p, a, bytes, s);
        dev_h2d(ctx->cur->vb.p, b, bytes, s);
        const unsigned T = 256, B = blocks_for(count, T);
        ZK_LAUNCH((k_to_mont<F>), dim3(B), dim3(T), 0, s, ptr<F>(ctx->cur->va), ptr<F>(ctx->cur->va), count);
        ZK_LAUNCH((k_to_mont<F>), dim3(B), dim3(T), 0, s, ptr<F>(ctx->cur->vb), ptr<F>(ctx->cur->vb), count);
        ZK_LAUNCH((k_field_op<F>), dim3(B), dim3(T), 0, s, ptr<F>(ctx->cur->va), ptr<F>(ctx->cur->vb), ptr<F>(ctx->cur->va), count, op);
        ZK_LAUNCH((k_from_mont<F>), dim3(B), dim3(T), 0, s, ptr<F>(ctx->cur->va), ptr<F>(ctx->cur->va), count);
        dev_d2h(out, ctx->cur->va.p, bytes, s);
        stream_sync(s);
    }

    static void r1cs_load(zkhip_ctx* ctx, zkhip_r1cs* cs, const u64* const rp[3], const u32* const col[3], const uint8_t* const val[3]) {
        Stream s = ctx->stream;
        std::vector<u64> long_rows, huge_rows;
        const u64 huge_min = env_int("ZKHIP_MATVEC_HUGE", 0, 1, 1) ? (u64)MATVEC_HUGE : ~(u64)0;     // (0: every long row to k_matvec_long, for A/B runs)
        for (int k = 0; k < 3; ++k) {
            const u64 nnz = rp[k][cs->n];
            require(rp[k][0] == 0, ZKHIP_ERR_BAD_ARG, "rowptr[0] must be 0");
            for (u64 i = 0; i < cs->n; ++i) require(rp[k][i] <= rp[k][i + 1], ZKHIP_ERR_BAD_ARG, "rowptr not monotone");
            cs->nnz_short[k] = nnz;
            for (u64 i = 0; i < cs->n; ++i)
                if (rp[k][i + 1] - rp[k][i] > MATVEC_LONG) {
                    (rp[k][i + 1] - rp[k][i] > huge_min ? huge_rows : long_rows).push_back((u64)k << 32 | i);
                    cs->nnz_short[k] -= rp[k][i + 1] - rp[k][i];
                }
            for (u64 q = 0; q < nnz; ++q) require(col[k][q] < cs->l + cs->w, ZKHIP_ERR_BAD_ARG, "column index out of range");
            cs->nnz[k] = nnz;
            cs->rp[k].ensure((cs->n + 1) * 8);
            cs->col[k].ensure(std::max<u64>(nnz, 1) * 4);
            cs->val[k].ensure(std::max<u64>(nnz, 1) * 32);
            dev_h2d(cs->rp[k].p, rp[k], (cs->n + 1) * 8, s);
            if (nnz) {
                dev_h2d(cs->col[k].p, col[k], nnz * 4, s);
                dev_h2d(cs->val[k].p, val[k], nnz * 32, s);
                ZK_LAUNCH((k_to_mont<Fr>), dim3(blocks_for(nnz, 256)), dim3(256), 0, s, ptr<Fr>(cs->val[k]), ptr<Fr>(cs->val[k]), nnz);
            }
        }
        cs->n_huge = huge_rows.size();
        long_rows.insert(long_rows.begin(), huge_rows.begin(), huge_rows.end());     // the huge rows first
        cs->n_long = long_rows.size();
        if (cs->n_long) {
            cs->long_rows.ensure(cs->n_long * 8);
            dev_h2d(cs->long_rows.p, long_rows.data(), cs->n_long * 8, s);
        }
        stream_sync(s);
    }
};

}  // namespace zk

#include "setup.cuh"
#include "gm17.cuh"

// ------------------------------------------------------------------ per-curve entry points
namespace zk {
struct CurveOps {
    void (*pk_load)(zkhip_ctx*, const uint8_t*, size_t, zkhip_pk*);
    void (*pk_table_levels)(zkhip_ctx*, zkhip_pk*);
    void (*pk_bind)(zkhip_ctx*, zkhip_pk*, const zkhip_r1cs*);
    void (*pk_bind_check)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*);
    void (*bound_level0_from_file)(zkhip_ctx*, int scheme, const zkhip_r1cs*, const uint8_t*, size_t, std::vector<uint8_t>&, std::vector<uint8_t>&, u64 fp[2]);
    void (*install_bound_ranges)(zkhip_ctx*, zkhip_pk*, const zkhip_r1cs*, const uint8_t*, size_t, const uint8_t*, size_t, const u64 fp[2]);
    void (*install_bound)(zkhip_ctx*, zkhip_pk*, const void*, const void*, bool on_device, const u64 fp[2]);
    void (*r1cs_fingerprint)(zkhip_ctx*, const zkhip_r1cs*, u64 fp[2]);
    bool (*can_split)(const zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*);
    void (*split_begin)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const void*, const uint8_t*, const uint8_t*, int half);
    const void* (*split_half_ptr)(zkhip_ctx*, const zkhip_pk*, int half);
    void (*split_fetch)(zkhip_ctx*, const zkhip_pk*, const void*, int, Event);
    void (*split_fetch_host)(zkhip_ctx*, const zkhip_pk*, const uint8_t*);
    void (*split_half_out)(zkhip_ctx*, const zkhip_pk*, uint8_t*);
    void (*split_end_partial)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, uint8_t*, zkhip_timings*);
    void (*split_end_device_sums)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const void**, size_t*, const void**, size_t*, zkhip_timings*);
    void (*r1cs_load)(zkhip_ctx*, zkhip_r1cs*, const u64* const rp[3], const u32* const col[3], const uint8_t* const val[3]);
    void (*prove)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, zkhip_timings*);
    void (*prove_resident)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, void*, const uint8_t*, const uint8_t*, uint8_t*, zkhip_timings*);
    void (*prove_batch)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, u32, const uint8_t*, void* const*, const uint8_t*, uint8_t*, zkhip_timings*);
    void (*prove_partial)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const void*, const uint8_t*, const uint8_t*, uint8_t*,
                          zkhip_timings*);
    void (*combine)(const zkhip_pk*, u32, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*);
    size_t partial_bytes;
    void (*prove_device_sums)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const uint8_t*, const uint8_t*, const void**, size_t*,
                              const void**, size_t*, zkhip_timings*);
    void (*gm17_prove_device_sums)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const uint8_t*, const void**, size_t*, const void**,
                                   size_t*, zkhip_timings*);
    void (*record_from_sums)(zkhip_ctx*, const zkhip_pk*, const uint8_t*, const uint8_t*, uint8_t*);
    void (*assignment_upload)(zkhip_ctx*, zkhip_assignment*, const uint8_t*);
    void (*ntt)(zkhip_ctx*, u32, int, uint8_t*);
    void (*witness_map)(zkhip_ctx*, const zkhip_r1cs*, const uint8_t*, uint8_t*);
    void (*msm_g1)(zkhip_ctx*, u64, const uint8_t*, const uint8_t*, uint8_t*);
    void (*msm_g2)(zkhip_ctx*, u64, const uint8_t*, const uint8_t*, uint8_t*);
    void (*field_op)(zkhip_ctx*, int field, int op, u64, const uint8_t*, const uint8_t*, uint8_t*);
    void (*setup)(zkhip_ctx*, const zkhip_r1cs*, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, u64);
    // GM17 (gm17.cuh)
    void (*gm17_pk_load)(zkhip_ctx*, const uint8_t*, size_t, zkhip_pk*);
    void (*gm17_prove)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const void*, const uint8_t*, uint8_t*, zkhip_timings*);
    void (*gm17_prove_batch)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, u32, const uint8_t*, void* const*, const uint8_t*, uint8_t*,
                             zkhip_timings*);
    void (*gm17_setup)(zkhip_ctx*, const zkhip_r1cs*, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, u64);
    u64 (*gm17_key_bytes)(u64, u64, u64);
    void (*gm17_prove_partial)(zkhip_ctx*, const zkhip_pk*, const zkhip_r1cs*, const uint8_t*, const void*, const uint8_t*, uint8_t*, zkhip_timings*);
    void (*gm17_combine)(const zkhip_pk*, u32, const uint8_t*, const uint8_t*, uint8_t*);
    int (*ntt_log1)(zkhip_ctx*, int logN);   // the split of this context's NTT plan for a domain (what a key's h order depends on)
    bool (*msm_shape_ok)(const zkhip_ctx*, u64 n, int c, int sets);   // (c, sets) of a key usable as they are under this context's settings
    size_t packed_g1_bytes;  // size of one resident G1 base (G2: twice that): lets zkhip_pk_import validate an image's shape
    int fr_bits;             // scalar width: the number of table levels follows from it and the window width
};
template <class C>
static void field_op_dispatch(zkhip_ctx* ctx, int field, int op, u64 count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (field == 0) Prover<C>::template field_op_api<typename C::Fr>(ctx, op, count, a, b, out);
    else Prover<C>::template field_op_api<typename C::Fq>(ctx, op, count, a, b, out);
}
template <class C>
static int ntt_log1_of(zkhip_ctx* ctx, int logN) { return get_plan<C>(ctx, logN)->split(); }
template <class C>
static bool msm_shape_ok_of(const zkhip_ctx* ctx, u64 n, int c, int sets) {
    const MsmShape sh = msm_shape(ctx, n, C::Fr::Params::BITS, true, c, sets);
    return sh.c == c && (int)sh.sets == sets;
}
template <class C>
static CurveOps make_curve_ops() {
    CurveOps o;
    o.msm_shape_ok = &msm_shape_ok_of<C>;
    o.ntt_log1 = &ntt_log1_of<C>;
    o.pk_load = &PkLoader<C>::load;
    o.pk_table_levels = &PkLoader<C>::table_levels;
    o.pk_bind = &PkLoader<C>::bind;
    o.pk_bind_check = &PkLoader<C>::bind_check;
    o.bound_level0_from_file = &PkLoader<C>::bound_level0_from_file;
    o.install_bound_ranges = &PkLoader<C>::install_bound_ranges;
    o.install_bound = &PkLoader<C>::install_bound;
    o.r1cs_fingerprint = &PkLoader<C>::r1cs_fingerprint;
    o.can_split = &Prover<C>::can_split;
    o.split_begin = &Prover<C>::split_begin;
    o.split_half_ptr = &Prover<C>::split_half_ptr;
    o.split_fetch = &Prover<C>::split_fetch;
    o.split_fetch_host = &Prover<C>::split_fetch_host;
    o.split_half_out = &Prover<C>::split_half_out;
    o.split_end_partial = &Prover<C>::split_end_partial;
    o.split_end_device_sums = &Prover<C>::split_end_device_sums;
    o.r1cs_load = &Prover<C>::r1cs_load;
    o.prove = &Prover<C>::prove_host;
    o.prove_resident = &Prover<C>::prove_resident;
    o.prove_batch = &Prover<C>::prove_batch;
    o.prove_partial = &Prover<C>::prove_partial;
    o.combine = &Prover<C>::combine;
    o.partial_bytes = Prover<C>::PARTIAL_BYTES;
    o.prove_device_sums = &Prover<C>::prove_device_sums;
    o.gm17_prove_device_sums = &Gm17<C>::prove_device_sums;
    o.record_from_sums = &Prover<C>::record_from_sums;
    o.assignment_upload = &Prover<C>::assignment_upload;
    o.ntt = &Prover<C>::ntt_api;
    o.witness_map = &Prover<C>::witness_map_api;
    o.msm_g1 = &Prover<C>::template msm_api<typename C::Fq, 2>;
    o.msm_g2 = &Prover<C>::template msm_api<typename C::Fq2, 4>;
    o.field_op = &field_op_dispatch<C>;
    o.setup = &Setup<C>::run;
    o.gm17_pk_load = &Gm17<C>::load;
    o.gm17_prove = &Gm17<C>::prove;
    o.gm17_prove_batch = &Gm17<C>::prove_batch;
    o.gm17_setup = &Gm17<C>::setup;
    o.gm17_key_bytes = &Gm17<C>::key_bytes;
    o.gm17_prove_partial = &Gm17<C>::prove_partial;
    o.gm17_combine = &Gm17<C>::combine;
    o.packed_g1_bytes = packed_point_bytes<typename C::Fq>();
    o.fr_bits = C::Fr::Params::BITS;
    return o;
}
const CurveOps* curve_ops_bn254();
const CurveOps* curve_ops_bls381();
}  // namespace zk

/* zkhip.h — C ABI of libzkhip.so, the MI355X-native Groth16 proving backend.
 *
 * This is the drop-in boundary for ONE reference path: `Backend<T, G16>::generate_proof`
 *   trait      /root/reference/zokrates_proof_systems/src/lib.rs:98-112
 *   CPU impl   /root/reference/zokrates_ark/src/groth16.rs:20-53   (`impl Backend<T,G16> for Ark`)
 *   call site  /root/reference/zokrates_cli/src/ops/generate_proof.rs:187
 * A Rust crate `zokrates_hip` (source in INTEGRATION.md) implements that trait by calling the
 * functions below; everything above the trait (CLI, JSON, Solidity export, verify) is untouched.
 *
 * Conventions
 *  - Every function returns int32_t: 0 = ZKHIP_OK, < 0 = error class; text via zkhip_last_error().
 *    The library never throws across the boundary and never aborts.  (The reference panics on any
 *    failure — zokrates_ark/src/groth16.rs:41-44 `.unwrap()` — so the Rust shim turns non-zero
 *    into `panic!`.)
 *  - Field elements cross the boundary as canonical little-endian bytes (32 B for Fr; sz(Fq) =
 *    32 B for bn128, 48 B for bls12_381): the bytes of ark `ToBytes` / `Field::write`
 *    (/root/reference/zokrates_field/src/lib.rs:215-226).  Neither side needs the other's
 *    Montgomery constant.
 *  - Caller owns every input buffer for the duration of the call; the library copies what it
 *    keeps; outputs go to caller-allocated buffers; long-lived objects are opaque handles with
 *    paired create/free.  No callbacks.
 *  - A context is bound to one GPU and is not re-entrant (one call at a time per context);
 *    distinct contexts may be used from distinct threads.
 *  - There is no CPU fallback: without a usable gfx950 device zkhip_ctx_create fails with
 *    ZKHIP_ERR_DEVICE.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_OK 0
#define ZKHIP_ERR_BAD_ARG (-1)      /* null pointer, size mismatch, unsupported curve/size        */
#define ZKHIP_ERR_PARSE (-2)        /* malformed proving key / non-canonical field element         */
#define ZKHIP_ERR_NOMEM (-3)        /* host or device allocation failed                            */
#define ZKHIP_ERR_DEVICE (-4)       /* no GPU, HIP runtime error, kernel launch failure            */
#define ZKHIP_ERR_UNSATISFIED (-5)  /* assignment does not satisfy the R1CS (only when checked)    */

/* curve ids; names are zokrates_common::constants (/root/reference/zokrates_common/src/constants.rs:5-9) */
#define ZKHIP_CURVE_BN128 0
#define ZKHIP_CURVE_BLS12_381 1

typedef struct zkhip_ctx zkhip_ctx;
typedef struct zkhip_pk zkhip_pk;
typedef struct zkhip_r1cs zkhip_r1cs;
typedef struct zkhip_assignment zkhip_assignment;

/* Per-proof phase timings in milliseconds (HIP events on the library's own streams).  The MSMs run on their own
 * streams concurrently with each other and with the NTT pipeline, so the phase intervals overlap: they do not
 * add up to total_ms.
 * Replaces nothing in the reference (it has no prover timers) — added observability. */
typedef struct zkhip_timings {
    float h2d_ms;       /* assignment upload + Montgomery conversion                */
    float matvec_ms;    /* K1: sparse A,B,C mat-vec                                  */
    float ntt_ms;       /* K2-K4: 7 transforms + pointwise quotient                  */
    float msm_h_ms;     /* K5: h_query MSM (scalar prep + buckets + reduce)          */
    float msm_z_ms;     /* K6-K8: a/b_g1/l (G1) and b_g2 (G2) MSMs over z            */
    float finish_ms;    /* K9: window combine, assembly, to-affine (host)            */
    float total_ms;     /* wall clock of the whole call                              */
    float kernel_msm_accum_g1_ms; /* sum of G1 bucket-accumulation kernel time       */
    float kernel_msm_accum_g2_ms; /* G2 bucket-accumulation kernel time              */
    float kernel_ntt_ms;          /* the transform passes + quotient kernel, first launch to last (events) */
    float reserved[6];
} zkhip_timings;

/* ---- process-wide start options ---- */
/* The library keeps ~20 HIP streams busy per context (a stream per proof slot and MSM, the transform pipeline, staging, copy-out);
 * the HIP runtime multiplexes them onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of streams that share a queue
 * serialise: 8 queues measured +4 % proofs/s over 4, 16 another +1.5-2 %.  The runtime reads that setting ONCE, when it
 * initialises (the first HIP call of the process) — so it is the HOST's decision, made explicitly here and nowhere else: the
 * library itself never touches the process environment (rounds 1-5 did, in a load-time constructor: a drop-in loaded into someone
 * else's process must not change HIP's behaviour for code that is not its own).
 *   hw_queues > 0: ask the runtime for that many queues — GPU_MAX_HW_QUEUES is set unless the process has set it already;
 *                  effective only BEFORE the first HIP call of the process (ZKHIP_ERR_BAD_ARG once this library has made one).
 *                  8 is safe for any process; 16 only for a process with ONE resident prover (every queue reserves scratch
 *                  for the largest kernel frame launched on it: a process with many contexts ran out at 16).
 *   hw_queues = 0: nothing is changed; the call only reports (returns ZKHIP_OK).
 * Not calling zkhip_init at all is legal: the runtime's own default applies.  Not thread-safe against concurrent getenv in other
 * threads (setenv never is): call it from the thread that starts the process's GPU work, before the others exist. */
int32_t zkhip_init(int32_t hw_queues);

/* ---- context ---- */
/* Number of usable HIP devices (0 if none / no runtime). */
int32_t zkhip_device_count(void);
/* PCI address of HIP device `device` as "dddd:bb:dd.f" (NUL-terminated, at most cap bytes): what ties a device ordinal to the
 * host's view of the same GPU — /sys/bus/pci/devices/<address>/numa_node for the NUMA node a rank's host thread should run
 * on, .../hwmon for clocks and power under load.  No context needed. */
int32_t zkhip_device_pci_bus_id(int32_t device, char* out, size_t cap);
/* Create a context on HIP device `device`. */
int32_t zkhip_ctx_create(int32_t device, zkhip_ctx** out);
void zkhip_ctx_free(zkhip_ctx* ctx);
/* Last error text of this context (or of the failed create when ctx == NULL). Never NULL. */
const char* zkhip_last_error(const zkhip_ctx* ctx);
/* Measurement aid: the shader clock the device actually runs at over the next `duration_us` microseconds, in GHz — ONE wavefront
 * (a few registers, asleep between its two readings) compares the shader-cycle counter with the constant-rate wall clock, on a
 * stream of its own, so that it can run BESIDE whatever else the device is doing: called from a second host thread on a second
 * context of the same device while the first proves, it reads the clock the proving kernels run at (bench.py prices the
 * accumulation's VALU issue limit with it instead of a clock derived from counters).  Blocks for the duration. */
int32_t zkhip_ctx_clock_probe(zkhip_ctx* ctx, uint32_t duration_us, double* ghz_out);

/* Development / measurement knobs of one context (none changes a result; all are plain fields read by the host code,
 * nothing consults the environment after zkhip_ctx_create).  ZKHIP_TUNE_MSM_C: window width of the MSM tables built by
 * later key loads and of later ad-hoc MSMs (0 = automatic); _MSM_WAVES: accumulation waves per SIMD (0 = per point
 * type); _MSM_LANES: number of slices the sorted list is cut into (0 = one per resident work-item); _MSM_MIN_SLICE: the
 * finest cut; _FOLD_SCAN: 0 selects the double-and-add form of the last fold step; _SERIAL: 1 puts every kernel on one
 * stream (un-overlapped per-kernel timing); _NTT_SINGLE_MAX_LOG: largest domain transformed in a single pass;
 * _NTT_COLS: adjacent columns per workgroup of the NTT cols pass; _SLOTS: proofs in flight in the batch calls (1..4);
 * _Z_GATE: which accumulations over the assignment wait for the witness map of their proof (0 none, 1 the three G1
 * lanes — the default —, 2 the G2 lane as well); _FUSE_Z: 0 runs the three G1 MSMs over the assignment as separate
 * launches instead of one; _MSM_FUSED_WAVES: accumulation waves per SIMD of that one launch (0 = per point type);
 * _STREAM_JITTER: race-hunting mode, process-wide — before every kernel launch, copy and fill the library enqueues, a
 * spin kernel of random length (0 .. value microseconds, value <= 5000; 0 = off) is put on the same stream with
 * probability 1/2, so that the relative timing of the library's streams differs from call to call; an ordering that is
 * not carried by an event then shows up as a wrong result instead of passing by luck (tests/test_stream_jitter.py). */
#define ZKHIP_TUNE_MSM_C 1
#define ZKHIP_TUNE_MSM_WAVES 2
#define ZKHIP_TUNE_MSM_LANES 3
#define ZKHIP_TUNE_MSM_MIN_SLICE 4
#define ZKHIP_TUNE_FOLD_SCAN 5
#define ZKHIP_TUNE_SERIAL 6
#define ZKHIP_TUNE_NTT_SINGLE_MAX_LOG 7
#define ZKHIP_TUNE_NTT_COLS 8
#define ZKHIP_TUNE_SLOTS 9
#define ZKHIP_TUNE_Z_GATE 10
#define ZKHIP_TUNE_FUSE_Z 11
#define ZKHIP_TUNE_MSM_FUSED_WAVES 12
#define ZKHIP_TUNE_STREAM_JITTER 13
/* (14, 15: the knobs of the wide-window sort and fold round 3 built and round 4 removed) */
#define ZKHIP_TUNE_MSM_SETS 17        /* bucket sets of the MSM tables built by later key loads: 1 = every window multiple of every base
                                      * (one bucket set per MSM), 2 = every second multiple (two sets) ...; 0 = automatic: 1 while the
                                      * tables fit the device, else the smallest power of two that does (keys above 2^24 constraints) */
#define ZKHIP_TUNE_SKIP_INF 18        /* how the bucket accumulation meets bases at infinity: 0 = per table, from a count made at key load
                                      * (a table with more than one such base in 2048 lets lanes sit them out, the others send the rare
                                      * one through the general code), 1 = lanes always sit them out, 2 = always the general code        */
#define ZKHIP_TUNE_B_SORT 19          /* keys loaded from now on: the MSMs over b_query (G2, and G1 of a Groth16 key) on a sorted list of
                                      * their own that leaves out the variables whose B bases are the point at infinity — 0 = when a tenth
                                      * of them are (real circuits: every variable that does not occur in B), 1 = always, 2 = never       */
#define ZKHIP_TUNE_LONE_SCHED 21      /* how a LONE proof (the single-proof entry points) is laid out on the device, bits: 1 = its G2 accumulation
                                      * takes one workgroup per CU at raised wave priority, so that the short kernels of the witness map and the
                                      * sorts find room beside it; 2 = its G1 lanes over the assignment also wait for the sort of h; 4 = its witness
                                      * map waits for the sort of the assignment.  Batches are untouched; no setting changes a result. */
#define ZKHIP_TUNE_NTT_SKEW_US 22     /* start skew (microseconds, 0 = off) of the first round of workgroups of every transform pass: co-resident
                                      * workgroups drift apart so that one's loads land under the other's butterflies (kernels_ntt.cuh NttSkew) */
#define ZKHIP_TUNE_SORT_TWO_LEVEL 23   /* 1 (default): the placement pass of the MSMs' counting sort in two levels (coarse bins of 256 buckets, then
                                      * tiles: line-sized runs, 1 KiB of LDS per workgroup); 0: the one-level pass behind a 128 KiB histogram */
#define ZKHIP_TUNE_FOLD_LINES 24       /* the row and the column sums of an MSM's bucket matrix in one launch, a workgroup per line (kernels_msm.cuh 5a'):
                                      * 0 (default) never (two launches: rows, then columns over the stored bucket values), 1 always, 2 for launches over one table */
#define ZKHIP_TUNE_PIPE_PLAN 28        /* 1: every stream of the context made at its first proof and placed on the chip's four dispatchers by plan (core.cuh
                                         make_pipe_streams, ZK_PIPE_PLAN_RESIDENT: a layout found by a local search over four workloads, tools/plan_search.py;
                                         16 streams: ask zkhip_init for 16 queues).  For long-lived provers: level or better than streams in order of first use on
                                         every measured workload (stdlib SHA-256 +8-9 % proofs/s, a lone dense 2^20 proof 0.3 ms sooner); a one-proof process
                                         pays 0.15 s for the streams.  0 (default): streams made as they are first used.  Chosen before the first proof */
#define ZKHIP_TUNE_FOLD_HOP 27         /* the fold chain of an MSM on a second stream of its lane: 0 never, 1 every lane, 2 the G2 lane only */
#define ZKHIP_TUNE_NTT_FUSE_FIRST 26   /* 1 (default): the first butterfly round of a transform pass on the elements as they are fetched; 0: through LDS like the others */
#define ZKHIP_TUNE_FOLD_HG 25          /* shares a column of the bucket matrix is cut into by the two-launch fold's column pass (a power of two <= 256) */
#define ZKHIP_TUNE_HEAVY_RUNS 20      /* 1 (default): the partials of a bucket spread over many slices (the ones of a witness of bits) are
                                      * first summed run by run by a kernel of their own; 0: by the one workgroup of the bucket's row      */
#define ZKHIP_TUNE_NTT_MAX_SUBLOG 16 /* log2 of the longest sub-transform of an NTT pass (2..11; default 11): domains above 2^(2 x this)
                                      * take three passes instead of two — a test hook to reach the three-pass path (domains above
                                      * 2^22) with small domains.  Keys loaded before a change must be reloaded (their h order)        */
int32_t zkhip_ctx_tune(zkhip_ctx* ctx, int32_t which, int32_t value);

/* ---- proving key ----
 * Replaces `ProvingKey::<E>::deserialize_unchecked(proving_key)` at
 * /root/reference/zokrates_ark/src/groth16.rs:40-42: `bytes` is exactly the `proving.key` file the
 * reference's `setup` writes (zokrates_ark/src/groth16.rs:97-98, ark `serialize_unchecked`:
 * vk{alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1[]}, beta_g1, delta_g1, a_query[],
 * b_g1_query[], b_g2_query[], h_query[], l_query[]; Vec = u64 LE length + elements; affine points
 * uncompressed, infinity = bit 6 of the last byte).  Points are uploaded, converted to Montgomery
 * form and laid out for the MSM kernels; like the reference ("unchecked") no on-curve or subgroup
 * check is made. */
int32_t zkhip_pk_load_g16(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, zkhip_pk** out);
void zkhip_pk_free(zkhip_pk* pk);
/* Shape of a loaded key: out[0]=m (a_query len), out[1]=N-1 (h_query len), out[2]=w (l_query len),
 * out[3]=l (gamma_abc len). */
int32_t zkhip_pk_dims(const zkhip_pk* pk, uint64_t out[4]);

/* ---- constraint system ----
 * Replaces the `ConstraintSystem` that `Computation::generate_constraints`
 * (/root/reference/zokrates_ark/src/lib.rs:80-129) builds on every call: the three R1CS matrices in
 * CSR form, columns in ark variable order (column 0 = ONE, columns < l instance, then witness).
 * n = constraints, l = num_instance, w = num_witness.  val* = nnz x 32 B canonical LE. */
int32_t zkhip_r1cs_load(zkhip_ctx* ctx, int32_t curve, uint64_t n, uint64_t l, uint64_t w,
                        const uint64_t* rowptr_a, const uint32_t* col_a, const uint8_t* val_a,
                        const uint64_t* rowptr_b, const uint32_t* col_b, const uint8_t* val_b,
                        const uint64_t* rowptr_c, const uint32_t* col_c, const uint8_t* val_c,
                        zkhip_r1cs** out);
void zkhip_r1cs_free(zkhip_r1cs* r1cs);

/* ---- the hot path ----
 * Replaces `Groth16::<E>::prove(&pk, computation, rng)` at /root/reference/zokrates_ark/src/groth16.rs:44
 * ([UPSTREAM] ark_groth16::create_random_proof).
 *   z        : m x 32 B, full assignment in ark order (z[0] must be 1)
 *   r, s     : the two blinding scalars (32 B each).  They are inputs so that the Rust shim samples
 *              them from the caller's RNG exactly as ark does (`Fr::rand(rng)` twice, r then s).
 *   proof_out: 8 x sz(Fq) bytes  A.x A.y | B.x.c0 B.x.c1 B.y.c0 B.y.c1 | C.x C.y  (canonical LE),
 *              then 3 bytes: infinity flags of A, B, C  — the same bytes ark `ToBytes` gives
 *              `parse_g1`/`parse_g2` (/root/reference/zokrates_ark/src/lib.rs:150-218).
 *   timings  : optional. */
int32_t zkhip_prove_g16(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z,
                        const uint8_t* r, const uint8_t* s, uint8_t* proof_out, zkhip_timings* timings);

/* The same with the assignment already resident in HBM (a prover service uploads the witness of proof
 * k+1 while proof k runs; bench.py times this entry point so that the timed region starts with every
 * input in device memory).  zkhip_assignment_upload checks z[0] == 1 and copies m x 32 B to the GPU. */
int32_t zkhip_assignment_upload(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment** out);
void zkhip_assignment_free(zkhip_assignment* z);
int32_t zkhip_prove_g16_resident(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, zkhip_assignment* z,
                                 const uint8_t* r, const uint8_t* s, uint8_t* proof_out, zkhip_timings* timings);

/* Steady-state variant for proofs/sec: `count` assignments (each m x 32 B, contiguous), `count`
 * (r, s) pairs (64 B each) and `count` proof slots (8*sz(Fq)+3 B each).  Same results as `count`
 * calls of zkhip_prove_g16; two proofs are kept in flight, so the latency-bound tail of one proof
 * overlaps the MSMs of the next (timings: per-phase sums over the batch, total_ms = wall clock). */
int32_t zkhip_prove_g16_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count,
                              const uint8_t* z, const uint8_t* rs, uint8_t* proofs_out, zkhip_timings* timings);
/* The same over assignments already resident in HBM (zs[i] may repeat). */
int32_t zkhip_prove_g16_resident_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count,
                                       zkhip_assignment* const* zs, const uint8_t* rs, uint8_t* proofs_out,
                                       zkhip_timings* timings);

/* ---- one proof across several GPUs ----
 * Every MSM is a sum over independent (scalar, base) pairs, so rank k of `world` loads only its index range of
 * a_query / b_g1_query / b_g2_query / l_query / h_query (zkhip_pk_load_g16_shard), computes the partial sums of the five
 * MSMs over that range (zkhip_prove_g16_partial; the mat-vec / NTT stage is replicated on every rank, it is < 10 % of
 * the work) and the ranks exchange ONE fixed-size record each (zkhip_partial_size bytes: five points in XYZZ
 * coordinates) — an all-gather over RCCL in zokrates_amd/parallel.py.  zkhip_combine_g16 adds the records and
 * assembles the proof; the result is bit-identical to zkhip_prove_g16 with the whole key. */
int32_t zkhip_pk_load_g16_shard(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, uint32_t rank,
                                uint32_t world, zkhip_pk** out);
int32_t zkhip_partial_size(int32_t curve, uint64_t* bytes);
/* z: host assignment (m x 32 B) or NULL to use the resident one */
int32_t zkhip_prove_g16_partial(zkhip_ctx* ctx, const zkhip_pk* pk_shard, const zkhip_r1cs* r1cs, const uint8_t* z,
                                zkhip_assignment* z_resident, const uint8_t* r, const uint8_t* s, uint8_t* partial_out,
                                zkhip_timings* timings);
/* partials: count x zkhip_partial_size bytes, one record per rank, any order; pk: any shard (or the whole key) */
int32_t zkhip_combine_g16(zkhip_ctx* ctx, const zkhip_pk* pk, uint32_t count, const uint8_t* partials, const uint8_t* r,
                          const uint8_t* s, uint8_t* proof_out);

/* ---- primitives (exported for parity tests and micro-benchmarks) ---- */
/* [UPSTREAM] ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place (App. A.4).
 * data: 2^log_n x 32 B canonical LE, natural order in and out, transformed in place.
 * dir: 0 fft, 1 ifft, 2 coset_fft, 3 coset_ifft. */
int32_t zkhip_ntt(zkhip_ctx* ctx, int32_t curve, uint32_t log_n, int32_t dir, uint8_t* data);
/* LibsnarkReduction::witness_map (App. A.3): h coefficients (N x 32 B, natural order). */
int32_t zkhip_witness_map(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* z, uint8_t* h_out);
/* [UPSTREAM] ark_ec::msm::VariableBaseMSM::multi_scalar_mul (App. A.5).
 * bases: n affine points in the ark uncompressed encoding (2 x sz(Fq) for G1, 4 x sz(Fq) for G2);
 * scalars: n x 32 B canonical LE; out: affine coords canonical LE + 1 infinity-flag byte. */
int32_t zkhip_msm_g1(zkhip_ctx* ctx, int32_t curve, uint64_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out);
int32_t zkhip_msm_g2(zkhip_ctx* ctx, int32_t curve, uint64_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out);
/* Element-wise Montgomery field ops on the device (parity tests): op 0 add, 1 sub, 2 mul;
 * field 0 = Fr, 1 = Fq; a, b, out: count x sz(field) canonical LE. */
int32_t zkhip_field_op(zkhip_ctx* ctx, int32_t curve, int32_t field, int32_t op, uint64_t count,
                       const uint8_t* a, const uint8_t* b, uint8_t* out);

/* ---- "next" row N3: setup ----
 * Replaces `Groth16::<E>::circuit_specific_setup` at /root/reference/zokrates_ark/src/groth16.rs:95
 * ([UPSTREAM] generate_random_parameters, App. A.6) with the randomness made explicit:
 * toxic = alpha, beta, gamma, delta, tau (5 x 32 B); g1/g2 = group generators in ark uncompressed
 * encoding (ark samples random ones; NULL = the standard generators).  Writes the ark
 * `serialize_unchecked` proving key into pk_out (size from zkhip_setup_g16_size). */
int32_t zkhip_setup_g16_size(const zkhip_r1cs* r1cs, uint64_t* pk_bytes);
int32_t zkhip_setup_g16(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* toxic, const uint8_t* g1,
                        const uint8_t* g2, uint8_t* pk_out, uint64_t pk_cap);

/* ---- GM17: the second proof system behind the same Backend trait (BASELINE.json config 5) ----
 * Replaces `<Ark as Backend<T, GM17>>::generate_proof` (/root/reference/zokrates_ark/src/gm17.rs:43-78):
 * zkhip_pk_load_gm17 stands for `ProvingKey::deserialize_unchecked` (:60-62) — `bytes` is the `proving.key` written by
 * the reference's GM17 `setup` (:27-28; [UPSTREAM] ark_gm17::ProvingKey in `serialize_unchecked` layout: vk{h_g2,
 * g_alpha_g1, h_beta_g2, g_gamma_g1, h_gamma_g2, query[]}, a_query[], b_query[] (G2), c_query_1[], c_query_2[],
 * g_gamma_z, h_gamma_z (G2), g_ab_gamma_z, g_gamma2_z2, g_gamma2_z_t[]) — and zkhip_prove_gm17 for `GM17::prove`
 * (:64; [UPSTREAM] ark_gm17::create_random_proof).  The constraint system is the same zkhip_r1cs (the R1CS -> SAP
 * reduction happens on the device); the assignment is the same m x 32 B vector in ark order.
 *   d1_d2_r  : the three blinding scalars (3 x 32 B) in the order ark samples them (`Fr::rand(rng)` x 3: d1, d2, r)
 *   proof_out: as zkhip_prove_g16 (A | B | C canonical LE + 3 infinity flags); zkhip_pk_dims gives out[0] = SAP
 *              variables (a_query len), out[1] = g_gamma2_z_t len, out[2] = c_query_1 len, out[3] = query len. */
int32_t zkhip_pk_load_gm17(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, zkhip_pk** out);
int32_t zkhip_prove_gm17(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, const uint8_t* d1_d2_r,
                         uint8_t* proof_out, zkhip_timings* timings);
int32_t zkhip_prove_gm17_resident(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, zkhip_assignment* z,
                                  const uint8_t* d1_d2_r, uint8_t* proof_out, zkhip_timings* timings);
/* `count` proofs, two in flight (see zkhip_prove_g16_resident_batch); d1_d2_r: count x 96 B. */
int32_t zkhip_prove_gm17_resident_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count,
                                        zkhip_assignment* const* zs, const uint8_t* d1_d2_r, uint8_t* proofs_out,
                                        zkhip_timings* timings);
/* One GM17 proof across several GPUs, exactly as for Groth16 (zkhip_pk_load_g16_shard ...): rank k loads its index range of
 * a_query / b_query / c_query_1 / c_query_2 / g_gamma2_z_t, computes the five partial sums over it, the ranks exchange one
 * zkhip_partial_size-byte record each, zkhip_combine_gm17 adds them and assembles the proof (bit-identical). */
int32_t zkhip_pk_load_gm17_shard(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, uint32_t rank,
                                 uint32_t world, zkhip_pk** out);
int32_t zkhip_prove_gm17_partial(zkhip_ctx* ctx, const zkhip_pk* pk_shard, const zkhip_r1cs* r1cs, const uint8_t* z,
                                 zkhip_assignment* z_resident, const uint8_t* d1_d2_r, uint8_t* partial_out,
                                 zkhip_timings* timings);
int32_t zkhip_combine_gm17(zkhip_ctx* ctx, const zkhip_pk* pk, uint32_t count, const uint8_t* partials,
                           const uint8_t* d1_d2_r, uint8_t* proof_out);
/* Replaces `GM17::circuit_specific_setup` at /root/reference/zokrates_ark/src/gm17.rs:25 ([UPSTREAM]
 * ark_gm17::generate_parameters) with the randomness explicit: toxic = alpha, beta, gamma, t (4 x 32 B; ark's
 * generate_random_parameters fixes gamma = 1); g1/g2 as in zkhip_setup_g16. */
int32_t zkhip_setup_gm17_size(const zkhip_r1cs* r1cs, uint64_t* pk_bytes);
int32_t zkhip_setup_gm17(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* toxic, const uint8_t* g1,
                         const uint8_t* g2, uint8_t* pk_out, uint64_t pk_cap);

/* ---- one proof across several GPUs of ONE process (§8b "multi-GPU handled inside one context") ----
 * A caller behind the reference's trait (`Backend::generate_proof`, /root/reference/zokrates_proof_systems/src/lib.rs:98-112,
 * call site /root/reference/zokrates_cli/src/ops/generate_proof.rs:187) is a single process without a collective
 * library; zkhip_multi gives it the latency mode of the sharded entry points above without one.  It owns one context
 * per listed device (a device may be listed more than once: the members then share it, which is how the path is tested
 * on a one-GPU box), member k holds shard k of n of the proving key (1/n of every base table) and a replica of the
 * constraint system.  zkhip_prove_*_multi runs the members' shares on one host thread each — assignment upload,
 * replicated mat-vec / NTT stage, the five MSMs over the member's index ranges — collects the n canonical partial
 * records (zkhip_partial_size bytes each, produced in host memory: nothing larger ever has to cross between devices)
 * and assembles the proof on the calling thread.  The result is bit-identical to the single-GPU proof.
 * zkhip_multi_ctx lends a member's context (e.g. to run zkhip_setup_g16 on member 0); errors of the zkhip_multi_* /
 * zkhip_prove_*_multi calls are reported by zkhip_multi_last_error. */
typedef struct zkhip_multi zkhip_multi;
int32_t zkhip_ctx_create_multi(const int32_t* devices, int32_t n, zkhip_multi** out);
void zkhip_multi_free(zkhip_multi* m);
int32_t zkhip_multi_size(const zkhip_multi* m);
zkhip_ctx* zkhip_multi_ctx(zkhip_multi* m, int32_t member);
/* The exchange step of zkhip_prove_*_multi over RCCL instead of host memory (BASELINE.json's "RCCL ...
 * over xGMI"): on != 0 creates one RCCL communicator per member (ncclCommInitAll; librccl.so.1 is bound with dlopen at
 * this call, libzkhip does not link it) and from then on every member leaves the bucket-set sums of its five partial
 * MSMs on its device and the members ncclAllGather them (two small messages per proof, latency-bound); member 0's copy
 * is read back and combined into the same proof bytes.  Needs distinct devices (ZKHIP_ERR_BAD_ARG otherwise: RCCL
 * refuses two ranks on one GPU) and a loadable librccl (ZKHIP_ERR_DEVICE); on failure the host exchange stays in use.
 * zkhip_multi_exchange describes the exchange in use (RCCL version and rank count, or the host path). */
int32_t zkhip_multi_use_rccl(zkhip_multi* m, int32_t on);
const char* zkhip_multi_exchange(const zkhip_multi* m);
const char* zkhip_multi_last_error(const zkhip_multi* m);
int32_t zkhip_multi_r1cs_load(zkhip_multi* m, int32_t curve, uint64_t n, uint64_t l, uint64_t w, const uint64_t* rowptr_a,
                              const uint32_t* col_a, const uint8_t* val_a, const uint64_t* rowptr_b, const uint32_t* col_b,
                              const uint8_t* val_b, const uint64_t* rowptr_c, const uint32_t* col_c, const uint8_t* val_c);
int32_t zkhip_multi_pk_load_g16(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len);
int32_t zkhip_multi_pk_load_gm17(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len);
int32_t zkhip_prove_g16_multi(zkhip_multi* m, const uint8_t* z, const uint8_t* r, const uint8_t* s, uint8_t* proof_out,
                              zkhip_timings* timings);
int32_t zkhip_prove_gm17_multi(zkhip_multi* m, const uint8_t* z, const uint8_t* d1_d2_r, uint8_t* proof_out,
                               zkhip_timings* timings);
/* Throughput mode of the same object (SURVEY.md §8e "replicas"): zkhip_multi_pk_load_g16_replicas puts the WHOLE key on every
 * member, zkhip_prove_g16_multi_batch splits `count` independent proofs (z: count x m x 32 B, rs: count x 64 B = r | s,
 * proofs_out: count x proof bytes) into one contiguous block per member and runs the members' pipelined batch calls on
 * one host thread each — no communication at all; what a long-lived prover service behind the trait would call. */
int32_t zkhip_multi_pk_load_g16_replicas(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len);
/* The members' keys (shards or replicas, Groth16 or GM17) bound to the members' constraint system — see zkhip_pk_bind_r1cs below;
 * `key_bytes` = the key file the members were loaded from.  All members or none. */
int32_t zkhip_multi_bind(zkhip_multi* m, const uint8_t* key_bytes, size_t len);
int32_t zkhip_multi_unbind(zkhip_multi* m);
/* The witness map of ONE proof split between the members (north_star: "NTT domain shard"): over keys bound to the
 * system a proof needs a and b on the coset and nothing else of the witness map, so members of even rank transform a, members of odd
 * rank b (two transforms each instead of four), partners copy each other's vector — N x 32 bytes device to device (xGMI peer copy
 * between GPUs) — and every member forms the products of its own index range.  On by default for Groth16 members that are bound
 * (zkhip_multi_bind) over domains of at least 2^18 (ZKHIP_SPLIT_MIN_LOG; below, two small transforms cost less than the exchange);
 * every other case runs the whole map on every member, as SURVEY.md §8e recommends for it.  Same proof bytes either way.
 * zkhip_multi_transform_split(m, 0 / 1) switches it (-1: leave it), returning the previous setting; zkhip_multi_last_split tells
 * whether the last zkhip_prove_*_multi did split. */
int32_t zkhip_multi_transform_split(zkhip_multi* m, int32_t on);
int32_t zkhip_multi_last_split(const zkhip_multi* m);
int32_t zkhip_prove_g16_multi_batch(zkhip_multi* m, uint32_t count, const uint8_t* z, const uint8_t* rs, uint8_t* proofs_out,
                                    zkhip_timings* timings);

/* ---- "next" row N2: proving-key cache ----
 * zkhip_pk_export writes the *resident* form of a loaded key (Groth16 or GM17, whole or one shard): packed Montgomery
 * points, MSM-ready order, the extended base vectors — the bytes the GPU holds, plus a small header.
 * zkhip_pk_import brings such an image back with five (a bound key: seven) host-to-device copies and no parsing or conversion, taking
 * `ProvingKey::deserialize_unchecked` (/root/reference/zokrates_ark/src/groth16.rs:40-42; one Fq multiplication per
 * coordinate and a single-threaded read in the reference) off the per-invocation path.
 * A resident key is a set of five MSM tables: level 0 = the key's points, the further levels their precomputed window
 * multiples.  The image carries level 0 only and the import recomputes the other levels on the device
 * (~0.1 s for a 2^20-constraint key — less than reading them from a disk would take: an image that carried every level was
 * measured 2x slower end to end in round 3 and has been removed).
 * The library does no file I/O: the caller stores the image wherever it likes, keyed e.g. by the SHA-256 of the
 * `proving.key` it came from (`zkhip-cli generate-proof --key-cache DIR` does exactly that).  An image
 * is tied to the library build that wrote it (magic + layout version); a foreign image is rejected with ZKHIP_ERR_PARSE. */
int32_t zkhip_pk_export_size(const zkhip_pk* pk, uint64_t* bytes);
int32_t zkhip_pk_export(const zkhip_pk* pk, uint8_t* out, uint64_t cap);
int32_t zkhip_pk_import(zkhip_ctx* ctx, const uint8_t* bytes, size_t len, zkhip_pk** out);

/* ---- a resident prover's key: bound to its constraint system ----
 * The reference turns the evaluations of a, b, c into the coefficients of h = (ab - c)/Z with seven transforms per proof
 * ([UPSTREAM] ark-groth16 0.3.0 `LibsnarkReduction::witness_map`, reached from `Groth16::prove`,
 * /root/reference/zokrates_ark/src/groth16.rs:44), only to pair the coefficients with `h_query`.  Those transforms are linear,
 * so a prover that keeps a key resident can apply them to the key's BASES once: zkhip_pk_bind_r1cs computes
 *   H'_j = sum_i (g^-i / N) w^(-ij) h_query[i]           (pairs with a(g w^j) b(g w^j) / Z(g), the quotient's evaluations) and
 *   L'_v = l_query[v] + sum_k C[k][v] H''_k              (c's inverse transform and its mat-vec, folded into the l bases)
 * and keeps them beside the key's own tables.  From then on zkhip_prove_g16* with THIS key and THIS constraint system takes four
 * transforms and the mat-vec of A and B only; the proof's group elements — hence its bytes — are the same (the same holds for an
 * assignment that does not satisfy the system: what the reference's MSM over h[..N-1] ignores, the bound bases ignore).  Any
 * other constraint system, and every call after zkhip_pk_unbind, takes the key's own tables.  Costs two size-N transforms
 * over G1 points (about a second at 2^20) and two more MSM tables of device memory (ZKHIP_ERR_NOMEM when they do not fit; the key stays
 * usable, unbound).  A call that is refused for its arguments leaves an earlier binding untouched.
 * GM17 keys bind the same way (/root/reference/zokrates_ark/src/gm17.rs:63: the SAP quotient (U^2 - W)/Z is linear in W and in the
 * last transform): TWO transforms per proof instead of four on the twice-larger SAP domain, W's share on the c_query_1 bases.
 * A SHARD of a multi-GPU key holds only its index ranges of the bases, and the transforms need every base once:
 * zkhip_pk_bind_r1cs_shard takes the key FILE again (the bytes the shard was loaded from), computes H' / L' over the whole range
 * and keeps this shard's ranges (works for a whole key as well); zkhip_multi_bind does it for the members of a zkhip_multi — one
 * member computes, every member installs its ranges.
 * A key image (zkhip_pk_export) of a bound key carries level 0 of H' / L' and a fingerprint of the constraint system
 * (zkhip_r1cs_fingerprint: a position-keyed checksum of the resident matrices — a guard against mix-ups, not a commitment);
 * zkhip_pk_bind_r1cs on the imported key attaches the tables when the fingerprint of `r1cs` agrees — a restart costs a read and a
 * checksum, not the transforms.
 * zkhip_pk_is_bound: 1 if proofs over `r1cs` would take the bound tables, else 0. */
int32_t zkhip_pk_bind_r1cs(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* r1cs);
int32_t zkhip_pk_unbind(zkhip_pk* pk);
int32_t zkhip_pk_bind_r1cs_shard(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* key_bytes, size_t len);
/* The same split for ranks that are separate PROCESSES (one per GPU, torch.distributed / MPI around them): `begin` stages the
 * assignment, starts this rank's MSMs over it and transforms ITS half of the witness map — half = 0: a, 1: b on the coset — into
 * `half_out` (N x 32 bytes of host memory; N = the key's domain); the caller exchanges halves with a rank of the other parity;
 * `end` takes the partner's half and leaves the rank's partial record (as zkhip_prove_g16_partial).  The key must be bound
 * (zkhip_pk_bind_r1cs_shard).  zokrates_amd/parallel.py prove_sharded does exactly this over RCCL / gloo. */
int32_t zkhip_prove_g16_split_begin(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment* z_resident,
                                    const uint8_t* r, const uint8_t* s, int32_t half, uint8_t* half_out);
int32_t zkhip_prove_g16_split_end(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* other_half, uint8_t* partial_out,
                                  zkhip_timings* timings);
int32_t zkhip_r1cs_fingerprint(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, uint64_t out[2]);
int32_t zkhip_pk_is_bound(const zkhip_pk* pk, const zkhip_r1cs* r1cs);

/* ---- "next" row N1: ZoKrates' own input files (host only: no context, no device work) ----
 * zkhip_prog_parse replaces `ProgEnum::deserialize` (/root/reference/zokrates_ast/src/ir/serialize.rs:306-390: header,
 * sections, per-statement CBOR) followed by `Computation::generate_constraints`
 * (/root/reference/zokrates_ark/src/lib.rs:80-129): `bytes` is the `out` file `zokrates compile` writes; the result is
 * the R1CS in ark variable order — ONE, then instance variables (public arguments in argument order, `~out_k` as
 * first seen), then witness variables (private arguments, then the others as first seen walking the A, B, C linear
 * combinations of every constraint); duplicate variables in a combination are summed, zero coefficients dropped;
 * directives and logs are ignored.  Errors are reported through zkhip_last_error(NULL).
 *   zkhip_prog_dims: out[0] curve id, [1] n, [2] l, [3] w, [4] return count, [5] public arguments, [6] nnz(A)+nnz(B)+nnz(C)
 *   zkhip_prog_matrix: CSR arrays of A (0), B (1), C (2), valid until zkhip_prog_free — the arguments of zkhip_r1cs_load
 *   zkhip_prog_variable_order: the ZoKrates variable id of every column (0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1})
 *   zkhip_prog_r1cs_load: zkhip_r1cs_load of those matrices
 * zkhip_prog_assignment replaces `Witness::read` (/root/reference/zokrates_ast/src/ir/witness.rs:55-71) and the
 * `witness.remove(..)` walk of generate_constraints: `witness` is the file `zokrates compute-witness` writes; z_out
 * (m x 32 B, may be NULL) is the assignment in ark order — the `z` of zkhip_prove_g16 — and inputs_out (may be NULL;
 * inputs_cap elements of 32 B) receives `public_inputs_values` (/root/reference/zokrates_ast/src/ir/mod.rs:278-288:
 * public arguments in argument order, then ~out_0, ~out_1, ...: the `inputs` of proof.json); a variable without a value
 * gives ZKHIP_ERR_UNSATISFIED (the reference panics with AssignmentMissing). */
typedef struct zkhip_prog zkhip_prog;
int32_t zkhip_prog_parse(const uint8_t* bytes, size_t len, zkhip_prog** out);
void zkhip_prog_free(zkhip_prog* prog);
int32_t zkhip_prog_dims(const zkhip_prog* prog, uint64_t out[8]);
int32_t zkhip_prog_matrix(const zkhip_prog* prog, int32_t which, const uint64_t** rowptr, const uint32_t** col,
                          const uint8_t** val);
int32_t zkhip_prog_variable_order(const zkhip_prog* prog, const int64_t** ids);
int32_t zkhip_prog_r1cs_load(zkhip_ctx* ctx, const zkhip_prog* prog, zkhip_r1cs** out);
/* The inverse of zkhip_prog_parse: an R1CS (CSR, canonical LE values, m columns) as the bytes of a ZoKrates `out` program
 * whose statements are exactly the constraints in row order — `ProgIterator::serialize`
 * (/root/reference/zokrates_ast/src/ir/serialize.rs:202-279) for a program without directives.  ids[j] is the ZoKrates
 * variable id of column j (0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1}); (arg_ids, arg_private)[n_args] is the argument
 * list.  A reader numbers variables in first-seen order, so the columns come back in this order only if the rows
 * mention them in it.  `out` must hold zkhip_prog_write_bound(n, nnz(A)+nnz(B)+nnz(C), n_args) bytes; *len = bytes
 * written.  Host only. */
int32_t zkhip_prog_write_bound(uint64_t n, uint64_t nnz, uint64_t n_args, uint64_t* bytes);
int32_t zkhip_prog_write(int32_t curve, uint64_t n, uint64_t m, const uint64_t* rowptr_a, const uint32_t* col_a, const uint8_t* val_a,
                         const uint64_t* rowptr_b, const uint32_t* col_b, const uint8_t* val_b, const uint64_t* rowptr_c,
                         const uint32_t* col_c, const uint8_t* val_c, const int64_t* ids, const int64_t* arg_ids,
                         const uint8_t* arg_private, uint64_t n_args, uint32_t return_count, uint8_t* out, uint64_t cap,
                         uint64_t* len);
int32_t zkhip_prog_assignment(const zkhip_prog* prog, const uint8_t* witness, size_t len, uint8_t* z_out,
                              uint8_t* inputs_out, uint64_t inputs_cap, uint64_t* n_inputs);

/* Library / device description, NUL-terminated, for logs. */
int32_t zkhip_describe(const zkhip_ctx* ctx, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */

// zkhip_api.hip — the C ABI of include/zkhip.h: argument checking, error mapping, per-curve dispatch.
// The drop-in boundary for `Backend<T, G16>::generate_proof` (/root/reference/zokrates_proof_systems/src/lib.rs:98-112).
#include "core.cuh"
#include "ingest.h"
#ifndef ZK_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded at run time, when zkhip_multi_use_rccl asks for it
#endif

// ------------------------------------------------------------------ C ABI
static const CurveOps* ops_for(int curve) {
    if (curve == ZKHIP_CURVE_BN128) return curve_ops_bn254();
    if (curve == ZKHIP_CURVE_BLS12_381) return curve_ops_bls381();
    throw ApiError{ZKHIP_ERR_BAD_ARG, "unknown curve id"};
}
// errors of calls that have no context (zkhip_ctx_create, zkhip_prog_*): one message per calling thread
static thread_local std::string g_create_err;

// after a failed call: let whatever was enqueued drain and mark the proof slots free, so the context stays usable
static void release_slots(zkhip_ctx* ctx) {
    bool any = false;
    for (auto& sl : ctx->slots) any = any || sl.busy;
    if (!any) return;
    dev_sync_all();
    for (auto& sl : ctx->slots) sl.busy = false;
}
template <class Fn>
static int32_t guarded(zkhip_ctx* ctx, Fn&& fn) {
    try {
        if (ctx) dev_set(ctx->device);
        fn();
        return ZKHIP_OK;
    } catch (const ApiError& e) {
        (ctx ? ctx->err : g_create_err) = e.msg;
        if (ctx) release_slots(ctx);
        return e.code;
    } catch (const DevError& e) {
        (ctx ? ctx->err : g_create_err) = e.msg;
        if (ctx) release_slots(ctx);
        return ZKHIP_ERR_DEVICE;
    } catch (const std::bad_alloc&) {
        (ctx ? ctx->err : g_create_err) = "out of host memory";
        if (ctx) release_slots(ctx);
        return ZKHIP_ERR_NOMEM;
    } catch (...) {
        (ctx ? ctx->err : g_create_err) = "unexpected internal error";
        if (ctx) release_slots(ctx);
        return ZKHIP_ERR_DEVICE;
    }
}

template <class Fn>
static int32_t guarded_host(Fn&& fn) {
    try {
        fn();
        return ZKHIP_OK;
    } catch (const IngestError& e) {
        g_create_err = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        g_create_err = "out of host memory";
        return ZKHIP_ERR_NOMEM;
    } catch (...) {
        g_create_err = "unexpected internal error";
        return ZKHIP_ERR_PARSE;
    }
}
// whether this library has made a HIP call in this process (after which GPU_MAX_HW_QUEUES is no longer read by the runtime)
static std::atomic<bool> g_hip_touched{false};
#ifndef ZK_EMU
// one wavefront: shader cycles against the constant-rate wall clock over `ticks` wall-clock ticks (zkhip_ctx_clock_probe)
static __global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - w0 < ticks) __builtin_amdgcn_s_sleep(64);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
#endif

extern "C" {

int32_t zkhip_init(int32_t hw_queues) {
    if (hw_queues < 0 || hw_queues > 64) { g_create_err = "hw_queues out of range (0 .. 64)"; return ZKHIP_ERR_BAD_ARG; }
    if (hw_queues == 0) return ZKHIP_OK;
#ifndef ZK_EMU
    if (g_hip_touched.load()) { g_create_err = "zkhip_init after the library's first HIP call: the runtime has read its settings already"; return ZKHIP_ERR_BAD_ARG; }
    char buf[16];
    snprintf(buf, sizeof(buf), "%d", (int)hw_queues);
    setenv("GPU_MAX_HW_QUEUES", buf, 0);      // (0: a value the process set itself stands)
#endif
    return ZKHIP_OK;
}

int32_t zkhip_device_count(void) { g_hip_touched.store(true); return dev_count(); }

int32_t zkhip_device_pci_bus_id(int32_t device, char* out, size_t cap) {
    if (!out || cap < 13) { g_create_err = "out is NULL or shorter than 13 bytes"; return ZKHIP_ERR_BAD_ARG; }
    out[0] = 0;
    return guarded(nullptr, [&] {
        require(device >= 0 && device < dev_count(), ZKHIP_ERR_BAD_ARG, "device index out of range");
        dev_pci_bus_id(device, out, cap);
    });
}

int32_t zkhip_ctx_create(int32_t device, zkhip_ctx** out) {
    if (!out) { g_create_err = "out is NULL"; return ZKHIP_ERR_BAD_ARG; }
    *out = nullptr;
    g_hip_touched.store(true);
    return guarded(nullptr, [&] {
        // ZKHIP_INIT_PROFILE=1: where the start of a process goes (stderr; a one-proof CLI run is mostly this)
        const bool prof = getenv("ZKHIP_INIT_PROFILE") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!prof) return;
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[zkhip init] %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
            t_last = t;
        };
        const int n = dev_count();
        lap("runtime start (device count)");
        require(n > 0, ZKHIP_ERR_DEVICE, "no HIP device available (libzkhip has no CPU fallback)");
        require(device >= 0 && device < n, ZKHIP_ERR_BAD_ARG, "device index out of range");
        dev_set(device);
        lap("set device");
        std::unique_ptr<zkhip_ctx> ctx(new zkhip_ctx());
        ctx->device = device;
        ctx->stream = stream_create_high_priority();
        lap("first stream");

        ctx->ws = ctx->stream;
        ctx->serial = getenv("ZKHIP_SERIAL") != nullptr;
        ctx->msm_c_env = env_int("ZKHIP_MSM_C", 2, MSM_MAX_C, 0);
        ctx->msm_sets = env_int("ZKHIP_MSM_SETS", 1, 64, 0);
        ctx->msm_waves = env_int("ZKHIP_MSM_WAVES", 1, 8, 0);
        ctx->ntt_single_max = env_int("ZKHIP_NTT_SINGLE_MAX_LOG", 0, NTT_MAX_SUBLOG, 10);
        ctx->ntt_max_sublog = env_int("ZKHIP_NTT_MAX_SUBLOG", 2, NTT_MAX_SUBLOG, NTT_MAX_SUBLOG);
        ctx->ntt_cols = env_int("ZKHIP_NTT_COLS", 1, 8, 2);
        if (ctx->ntt_cols & (ctx->ntt_cols - 1)) ctx->ntt_cols = 2;   // the cols pass has no tail handling: a power of two only
        ctx->nslots = env_int("ZKHIP_SLOTS", 1, ZK_NSLOTS, 3);
        ctx->z_gate = env_int("ZKHIP_Z_GATE", 0, 2, 1);
        ctx->g2_head_start = env_int("ZKHIP_G2_HEAD_START", 0, 2, 1);
        ctx->lone_sched = env_int("ZKHIP_LONE_SCHED", 0, 7, 0);
        ctx->split_min_log = env_int("ZKHIP_SPLIT_MIN_LOG", 0, 40, 18);
        ctx->ntt_skew_us = env_int("ZKHIP_NTT_SKEW_US", 0, 200, 0);
        ctx->fuse_z = env_int("ZKHIP_FUSE_Z", 0, 1, 1) != 0;
        ctx->heavy_runs = env_int("ZKHIP_MSM_HEAVY_RUNS", 0, 1, 1) != 0;
        { const int ht = env_int("ZKHIP_HEAVY_THREADS", 64, 256, 64); ctx->heavy_threads = ht >= 256 ? 256 : ht >= 128 ? 128 : 64; }
        { const int hg = env_int("ZKHIP_FOLD_HG", 1, 256, 32); ctx->fold_hg = 1 << ilog2_floor((u64)hg); }
        ctx->msm_fused_waves = env_int("ZKHIP_MSM_FUSED_WAVES", 1, 8, 0);
        ctx->msm_g1_waves = env_int("ZKHIP_MSM_G1_WAVES", 1, 16, 0);
        ctx->msm_g2_waves = env_int("ZKHIP_MSM_G2_WAVES", 1, 16, 0);
        ctx->sort_wgs = (u32)env_int("ZKHIP_SORT_WGS", 16, 4096, 256);
        ctx->sort_kh_log = env_int("ZKHIP_SORT_KH_LOG", 8, 15, 15);
        ctx->sort_two_level = env_int("ZKHIP_SORT_TWO_LEVEL", 0, 1, 1);
        ctx->fold_lines = env_int("ZKHIP_FOLD_LINES", 0, 2, 0);
        ctx->ntt_fuse_first = env_int("ZKHIP_NTT_FUSE_FIRST", 0, 1, 1);
        ctx->fold_hop = env_int("ZKHIP_FOLD_HOP", 0, 2, 0);
        ctx->stream_skew = env_int("ZKHIP_STREAM_SKEW", 0, 32, 0);
        if (const char* e = getenv("ZKHIP_PIPES")) ctx->pipe_plan = (e[0] == '-' || e[0] == '0') ? "" : (e[0] == '1' && !e[1]) ? ZK_PIPE_PLAN_RESIDENT : e;
        // every (slot, lane) has a stream of its own: with one stream per lane shared by the slots, the accumulation of
        // proof i+1 queued behind the latency-bound fold tail of proof i on the same lane (a kernel trace showed a lone
        // fold workgroup holding the machine 16 % of the time).  Slot 0 — the one single proofs and the primitives use — is
        // made now, the others when a batch call first needs them (slot_init: 20 ms of stream and event creation each, which a
        // one-proof process never spends).
        ctx->g2_first = env_int("ZKHIP_G2_PRIORITY", 0, 1, 1) != 0;
        slot_init(ctx.get(), ctx->slots[0]);
        lap("slot 0: events");
#ifdef ZK_EMU
        ctx->desc = "zkhip TEST EMULATOR (not a product build)";
#else
        hipDeviceProp_t prop;
        ZK_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        char buf[256];
        snprintf(buf, sizeof(buf), "zkhip on %s (%s), %d CUs, %.0f GiB", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                 prop.totalGlobalMem / 1073741824.0);
        ctx->desc = buf;
        ctx->cus = std::max(1, prop.multiProcessorCount);
        lap("device properties");
#endif
        *out = ctx.release();
    });
}
void zkhip_ctx_free(zkhip_ctx* ctx) {
    if (!ctx) return;
    try {
        dev_set(ctx->device);   // the calling thread may be bound to another device
    } catch (...) {
    }
    dev_sync_all();
    staging_drain();          // (the device's staging ring may still track transfers recorded on this context's streams)
    for (auto& sl : ctx->slots) {
        // (streams first, whether or not the slot was ever used: a stream plan makes the streams of every slot at the first proof)
        for (int k = 0; k < ZK_NLANES; ++k) {
            if (sl.lanes[k].made) stream_destroy(sl.lanes[k].stream);
            if (sl.lanes[k].fold_made && sl.lanes[k].fold_owned) stream_destroy(sl.lanes[k].fold_stream);
            if (sl.lanes[k].lone_fold_made) stream_destroy(sl.lanes[k].lone_fold_stream);
        }
        if (sl.fold_slot_made) stream_destroy(sl.fold_slot);
        if (!sl.ready) continue;
        for (auto& so : sl.sorts) event_destroy(so.ready);
        for (int k = 0; k < ZK_NLANES; ++k) {
            event_destroy(sl.lanes[k].done);
            event_destroy(sl.lanes[k].acc_done);
            event_destroy(sl.acc_b[k]);
            event_destroy(sl.acc_e[k]);
        }
        for (auto& e : sl.ev) event_destroy(e);
        event_destroy(sl.ntt_b);
        event_destroy(sl.ntt_e);
        event_destroy(sl.g1_go);
        host_free_pinned(sl.h_ws);
    }
    for (Stream q : ctx->skew_streams) stream_destroy(q);
    for (int t = 0; t < 3; ++t) if (ctx->fold_shared_made[t]) stream_destroy(ctx->fold_shared[t]);
    if (ctx->ntt_lone_made) stream_destroy(ctx->ntt_lone_stream);
    if (ctx->out_made) stream_destroy(ctx->out_stream);
    if (ctx->ntt_made) stream_destroy(ctx->ntt_stream);
    stream_destroy(ctx->stream);
    delete ctx;
}
int32_t zkhip_ctx_clock_probe(zkhip_ctx* ctx, uint32_t duration_us, double* ghz_out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(ghz_out && duration_us >= 1 && duration_us <= 10000000, ZKHIP_ERR_BAD_ARG, "null output or duration out of range (1 us .. 10 s)");
        *ghz_out = 0.0;
#ifndef ZK_EMU
        int khz = 0;
        ZK_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
        require(khz > 0, ZKHIP_ERR_DEVICE, "the device reports no wall-clock rate");
        unsigned long long* h = (unsigned long long*)host_alloc_pinned(16);
        h[0] = h[1] = 0;
        DBuf d;
        d.ensure(16);
        Stream s = stream_create_high_priority();      // (a stream of its own: it runs beside the context's proofs, not behind them)
        hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, (unsigned long long)duration_us * (unsigned long long)khz / 1000ull, (unsigned long long*)d.p);
        ZK_HIP_CHECK(hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, s));
        stream_sync(s);
        stream_destroy(s);
        if (h[1]) *ghz_out = (double)h[0] / ((double)h[1] / ((double)khz * 1e3)) / 1e9;
        host_free_pinned(h);
#endif
    });
}
int32_t zkhip_ctx_tune(zkhip_ctx* ctx, int32_t which, int32_t value) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        auto in = [&](int lo, int hi) { require(value >= lo && value <= hi, ZKHIP_ERR_BAD_ARG, "tunable value out of range"); };
        switch (which) {
            case ZKHIP_TUNE_MSM_C: if (value) in(2, MSM_MAX_C); ctx->msm_c_env = value; break;
            case ZKHIP_TUNE_MSM_SETS: in(0, 64); ctx->msm_sets = value; break;
            case ZKHIP_TUNE_SKIP_INF: in(0, 2); ctx->skip_inf_mode = value; break;
            case ZKHIP_TUNE_B_SORT: in(0, 2); ctx->b_sort_mode = value; break;
            case ZKHIP_TUNE_HEAVY_RUNS: in(0, 1); ctx->heavy_runs = value != 0; break;
            case ZKHIP_TUNE_MSM_WAVES: in(0, 8); ctx->msm_waves = value; break;
            case ZKHIP_TUNE_MSM_LANES: in(0, 1 << 24); ctx->msm_lanes = (u32)value; break;
            case ZKHIP_TUNE_MSM_MIN_SLICE: in(1, 1 << 20); ctx->msm_min_slice = (u32)value; break;
            case ZKHIP_TUNE_FOLD_SCAN: in(0, 1); ctx->fold_scan = value != 0; break;
            case ZKHIP_TUNE_SERIAL: in(0, 1); dev_sync_all(); ctx->serial = value != 0; break;
            case ZKHIP_TUNE_NTT_SINGLE_MAX_LOG: in(0, NTT_MAX_SUBLOG); dev_sync_all(); ctx->ntt_single_max = value; ctx->plans.clear(); break;
            case ZKHIP_TUNE_NTT_MAX_SUBLOG: in(2, NTT_MAX_SUBLOG); dev_sync_all(); ctx->ntt_max_sublog = value; ctx->plans.clear(); break;
            case ZKHIP_TUNE_SLOTS: in(1, ZK_NSLOTS); dev_sync_all(); ctx->nslots = value; break;
            case ZKHIP_TUNE_Z_GATE: in(0, 2); ctx->z_gate = value; break;
            case ZKHIP_TUNE_LONE_SCHED: in(0, 7); ctx->lone_sched = value; break;
            case ZKHIP_TUNE_NTT_SKEW_US: in(0, 200); ctx->ntt_skew_us = value; break;
            case ZKHIP_TUNE_SORT_TWO_LEVEL: in(0, 1); ctx->sort_two_level = value; break;
            case ZKHIP_TUNE_FOLD_LINES: in(0, 2); ctx->fold_lines = value; break;
            case ZKHIP_TUNE_NTT_FUSE_FIRST: in(0, 1); ctx->ntt_fuse_first = value; break;
            case ZKHIP_TUNE_FOLD_HOP: in(0, 2); ctx->fold_hop = value; break;
            case ZKHIP_TUNE_PIPE_PLAN:
                in(0, 1);
                require(!ctx->pipes_made, ZKHIP_ERR_BAD_ARG, "the streams of this context exist already: the plan is chosen before its first proof");
                if (!getenv("ZKHIP_PIPES")) ctx->pipe_plan = value ? ZK_PIPE_PLAN_RESIDENT : "";
                break;
            case ZKHIP_TUNE_FOLD_HG: in(1, 256); ctx->fold_hg = 1 << ilog2_floor((u64)value); break;
            case ZKHIP_TUNE_FUSE_Z: in(0, 1); ctx->fuse_z = value != 0; break;
            case ZKHIP_TUNE_MSM_FUSED_WAVES: in(0, 8); ctx->msm_fused_waves = value; break;
            case ZKHIP_TUNE_STREAM_JITTER: in(0, 5000); dev_sync_all(); jitter_state().max_us.store(value); break;
            case ZKHIP_TUNE_NTT_COLS:
                in(1, 8);
                require((value & (value - 1)) == 0, ZKHIP_ERR_BAD_ARG, "NTT_COLS must be 1, 2, 4 or 8 (the cols pass covers N2 / cols workgroups exactly)");
                dev_sync_all(); ctx->ntt_cols = value; ctx->plans.clear(); break;
            default: throw ApiError{ZKHIP_ERR_BAD_ARG, "unknown tunable"};
        }
    });
}
const char* zkhip_last_error(const zkhip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int32_t zkhip_describe(const zkhip_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || !cap) return ZKHIP_ERR_BAD_ARG;
    snprintf(buf, cap, "%s", ctx->desc.c_str());
    return ZKHIP_OK;
}

int32_t zkhip_pk_load_g16(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, zkhip_pk** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(bytes && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        std::unique_ptr<zkhip_pk> pk(new zkhip_pk());
        pk->curve = curve;
        pk->ctx = ctx;
        ops_for(curve)->pk_load(ctx, bytes, len, pk.get());
        *out = pk.release();
    });
}
int32_t zkhip_pk_load_g16_shard(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, uint32_t rank, uint32_t world, zkhip_pk** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(bytes && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        require(world >= 1 && world <= 64 && rank < world, ZKHIP_ERR_BAD_ARG, "rank / world out of range (1 <= world <= 64)");
        std::unique_ptr<zkhip_pk> pk(new zkhip_pk());
        pk->curve = curve;
        pk->ctx = ctx;
        pk->rank = rank;
        pk->world = world;
        ops_for(curve)->pk_load(ctx, bytes, len, pk.get());
        *out = pk.release();
    });
}
void zkhip_pk_free(zkhip_pk* pk) { delete pk; }
int32_t zkhip_pk_bind_r1cs(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* r1cs) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    // the argument checks come first and never touch the key: a call that is refused leaves an earlier binding (seconds of work)
    // as it was; only a failure INSIDE the rebuild — which starts by dropping the old tables — leaves the key as loaded
    bool started = false;
    const int32_t rc = guarded(ctx, [&] {
        require(pk && r1cs, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        for (auto& sl : ctx->slots) require(!sl.busy, ZKHIP_ERR_BAD_ARG, "a proof is in flight in this context");
        ops_for(pk->curve)->pk_bind_check(ctx, pk, r1cs);
        started = true;
        ops_for(pk->curve)->pk_bind(ctx, pk, r1cs);
    });
    if (rc != ZKHIP_OK && pk && started) zkhip_pk_unbind(pk);   // whatever failed half-way: the key is as it was loaded
    return rc;
}
int32_t zkhip_pk_unbind(zkhip_pk* pk) {
    if (!pk) return ZKHIP_ERR_BAD_ARG;
    return guarded(pk->ctx, [&] {
        pk->bound_uid = 0;
        pk->bound_fp[0] = pk->bound_fp[1] = 0;
        pk->h_bound.release();
        pk->l_bound.release();
    });
}
// a shard of a multi-GPU key (or a whole key): the binding computed from the key FILE — the transforms need every base once — of
// which this key keeps its own index ranges
int32_t zkhip_pk_bind_r1cs_shard(zkhip_ctx* ctx, zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* key_bytes, size_t len) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    bool started = false;
    const int32_t rc = guarded(ctx, [&] {
        require(pk && r1cs && key_bytes, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        for (auto& sl : ctx->slots) require(!sl.busy, ZKHIP_ERR_BAD_ARG, "a proof is in flight in this context");
        const CurveOps* ops = ops_for(pk->curve);
        ops->pk_bind_check(ctx, pk, r1cs);
        std::vector<uint8_t> h_host, l_host;
        u64 fp[2];
        ops->bound_level0_from_file(ctx, pk->scheme, r1cs, key_bytes, len, h_host, l_host, fp);
        started = true;
        ops->install_bound_ranges(ctx, pk, r1cs, h_host.data(), h_host.size(), l_host.data(), l_host.size(), fp);
    });
    if (rc != ZKHIP_OK && pk && started) zkhip_pk_unbind(pk);
    return rc;
}
// One rank's share of a proof whose witness map is split between the ranks of a multi-PROCESS prover (one process per GPU): begin
// computes this rank's half — `half` = 0: a, 1: b on the coset, N x 32 bytes (R'-form, packed) into `half_out` (host memory) — and
// leaves the proof in flight; the ranks exchange halves (partners of the other parity); end takes the partner's and finishes.
int32_t zkhip_prove_g16_split_begin(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment* z_resident, const uint8_t* r,
                                    const uint8_t* s, int32_t half, uint8_t* half_out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && (z || z_resident) && r && s && half_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        if (!z) require(z_resident->ctx == ctx && z_resident->curve == pk->curve && z_resident->m == r1cs->l + r1cs->w, ZKHIP_ERR_BAD_ARG,
                        "assignment does not match the constraint system");
        for (auto& sl : ctx->slots) require(!sl.busy, ZKHIP_ERR_BAD_ARG, "a proof is in flight in this context");
        const CurveOps* ops = ops_for(pk->curve);
        ops->split_begin(ctx, pk, r1cs, z, z ? nullptr : z_resident->scalars.p, r, s, half);
        ops->split_half_out(ctx, pk, half_out);
    });
}
int32_t zkhip_prove_g16_split_end(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* other_half, uint8_t* partial_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && other_half && partial_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        const CurveOps* ops = ops_for(pk->curve);
        ops->split_fetch_host(ctx, pk, other_half);
        ops->split_end_partial(ctx, pk, r1cs, partial_out, timings);
    });
}
int32_t zkhip_r1cs_fingerprint(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, uint64_t out[2]) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(r1cs && out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "constraint system belongs to another context");
        u64 fp[2];
        ops_for(r1cs->curve)->r1cs_fingerprint(ctx, r1cs, fp);
        out[0] = fp[0]; out[1] = fp[1];
    });
}
int32_t zkhip_pk_is_bound(const zkhip_pk* pk, const zkhip_r1cs* r1cs) {
    if (!pk || !r1cs) return 0;
    return pk->bound_uid != 0 && pk->bound_uid == r1cs->uid ? 1 : 0;
}
int32_t zkhip_pk_dims(const zkhip_pk* pk, uint64_t out[4]) {
    if (!pk || !out) return ZKHIP_ERR_BAD_ARG;
    out[0] = pk->m; out[1] = pk->hlen; out[2] = pk->w; out[3] = pk->l;
    return ZKHIP_OK;
}

int32_t zkhip_r1cs_load(zkhip_ctx* ctx, int32_t curve, uint64_t n, uint64_t l, uint64_t w, const uint64_t* rowptr_a, const uint32_t* col_a,
                        const uint8_t* val_a, const uint64_t* rowptr_b, const uint32_t* col_b, const uint8_t* val_b, const uint64_t* rowptr_c,
                        const uint32_t* col_c, const uint8_t* val_c, zkhip_r1cs** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(out && rowptr_a && rowptr_b && rowptr_c, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        require(l >= 1, ZKHIP_ERR_BAD_ARG, "num_instance must be >= 1 (the constant ONE)");
        require(l + w + 2 < ((u64)1 << 31) && n < ((u64)1 << 31), ZKHIP_ERR_BAD_ARG, "constraint system too large");
        std::unique_ptr<zkhip_r1cs> cs(new zkhip_r1cs());
        cs->curve = curve; cs->ctx = ctx; cs->n = n; cs->l = l; cs->w = w;
        cs->logN = ilog2_ceil(n + l);
        cs->N = (u64)1 << cs->logN;
        const u64* rp[3] = {rowptr_a, rowptr_b, rowptr_c};
        const u32* col[3] = {col_a, col_b, col_c};
        const uint8_t* val[3] = {val_a, val_b, val_c};
        ops_for(curve)->r1cs_load(ctx, cs.get(), rp, col, val);
        *out = cs.release();
    });
}
void zkhip_r1cs_free(zkhip_r1cs* cs) { delete cs; }

int32_t zkhip_prove_g16(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, const uint8_t* r, const uint8_t* s,
                        uint8_t* proof_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && z && r && s && proof_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_g16_partial + zkhip_combine_g16");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        ops_for(pk->curve)->prove(ctx, pk, r1cs, z, r, s, proof_out, timings);
    });
}

int32_t zkhip_assignment_upload(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(r1cs && z && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        require(r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "constraint system belongs to another context");
        std::unique_ptr<zkhip_assignment> a(new zkhip_assignment());
        a->curve = r1cs->curve; a->ctx = ctx; a->m = r1cs->l + r1cs->w;
        ops_for(r1cs->curve)->assignment_upload(ctx, a.get(), z);
        *out = a.release();
    });
}
void zkhip_assignment_free(zkhip_assignment* a) { delete a; }

int32_t zkhip_prove_g16_resident(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, zkhip_assignment* z, const uint8_t* r,
                                 const uint8_t* s, uint8_t* proof_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && z && r && s && proof_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_g16_partial + zkhip_combine_g16");
        require(pk->ctx == ctx && r1cs->ctx == ctx && z->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        require(z->curve == pk->curve && z->m == pk->m, ZKHIP_ERR_BAD_ARG, "assignment does not match the proving key");
        ops_for(pk->curve)->prove_resident(ctx, pk, r1cs, z->scalars.p, r, s, proof_out, timings);
    });
}

int32_t zkhip_partial_size(int32_t curve, uint64_t* bytes) {
    if (!bytes) return ZKHIP_ERR_BAD_ARG;
    try {
        *bytes = ops_for(curve)->partial_bytes;
    } catch (...) {
        return ZKHIP_ERR_BAD_ARG;
    }
    return ZKHIP_OK;
}
int32_t zkhip_prove_g16_partial(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment* z_resident,
                                const uint8_t* r, const uint8_t* s, uint8_t* partial_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && (z || z_resident) && r && s && partial_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        if (!z) require(z_resident->ctx == ctx && z_resident->curve == pk->curve && z_resident->m == pk->m, ZKHIP_ERR_BAD_ARG,
                        "assignment does not match the proving key");
        ops_for(pk->curve)->prove_partial(ctx, pk, r1cs, z, z ? nullptr : z_resident->scalars.p, r, s, partial_out, timings);
    });
}
int32_t zkhip_combine_g16(zkhip_ctx* ctx, const zkhip_pk* pk, uint32_t count, const uint8_t* partials, const uint8_t* r, const uint8_t* s,
                          uint8_t* proof_out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && partials && r && s && proof_out && count >= 1, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->scheme == 0, ZKHIP_ERR_BAD_ARG, "this is a GM17 proving key");
        ops_for(pk->curve)->combine(pk, count, partials, r, s, proof_out);
    });
}

int32_t zkhip_prove_g16_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count, const uint8_t* z, const uint8_t* rs,
                              uint8_t* proofs_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && rs && proofs_out && (z || count == 0), ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_g16_partial + zkhip_combine_g16");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        ops_for(pk->curve)->prove_batch(ctx, pk, r1cs, count, z, nullptr, rs, proofs_out, timings);
    });
}
int32_t zkhip_prove_g16_resident_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count, zkhip_assignment* const* zs,
                                       const uint8_t* rs, uint8_t* proofs_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && rs && proofs_out && (zs || count == 0), ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_g16_partial + zkhip_combine_g16");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        std::vector<void*> dev(count);
        for (uint32_t i = 0; i < count; ++i) {
            require(zs[i] && zs[i]->ctx == ctx && zs[i]->curve == pk->curve && zs[i]->m == pk->m, ZKHIP_ERR_BAD_ARG,
                    "assignment does not match the proving key");
            dev[i] = zs[i]->scalars.p;
        }
        ops_for(pk->curve)->prove_batch(ctx, pk, r1cs, count, nullptr, dev.data(), rs, proofs_out, timings);
    });
}

int32_t zkhip_ntt(zkhip_ctx* ctx, int32_t curve, uint32_t log_n, int32_t dir, uint8_t* data) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(data && dir >= 0 && dir <= 3, ZKHIP_ERR_BAD_ARG, "bad argument");
        ops_for(curve)->ntt(ctx, log_n, dir, data);
    });
}
int32_t zkhip_witness_map(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* z, uint8_t* h_out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(r1cs && z && h_out, ZKHIP_ERR_BAD_ARG, "null argument");
        ops_for(r1cs->curve)->witness_map(ctx, r1cs, z, h_out);
    });
}
int32_t zkhip_msm_g1(zkhip_ctx* ctx, int32_t curve, uint64_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(out && (n == 0 || (bases && scalars)), ZKHIP_ERR_BAD_ARG, "null argument");
        ops_for(curve)->msm_g1(ctx, n, bases, scalars, out);
    });
}
int32_t zkhip_msm_g2(zkhip_ctx* ctx, int32_t curve, uint64_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(out && (n == 0 || (bases && scalars)), ZKHIP_ERR_BAD_ARG, "null argument");
        ops_for(curve)->msm_g2(ctx, n, bases, scalars, out);
    });
}
int32_t zkhip_field_op(zkhip_ctx* ctx, int32_t curve, int32_t field, int32_t op, uint64_t count, const uint8_t* a, const uint8_t* b,
                       uint8_t* out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(a && b && out && op >= 0 && op <= 2 && (field == 0 || field == 1), ZKHIP_ERR_BAD_ARG, "bad argument");
        ops_for(curve)->field_op(ctx, field, op, count, a, b, out);
    });
}

int32_t zkhip_setup_g16_size(const zkhip_r1cs* cs, uint64_t* pk_bytes) {
    if (!cs || !pk_bytes) return ZKHIP_ERR_BAD_ARG;
    const u64 fqb = cs->curve == ZKHIP_CURVE_BN128 ? 32 : 48;
    const u64 g1 = 2 * fqb, g2 = 4 * fqb, m = cs->l + cs->w;
    *pk_bytes = g1 + 3 * g2 + 8 + cs->l * g1 + 2 * g1 + 8 + m * g1 + 8 + m * g1 + 8 + m * g2 + 8 + (cs->N - 1) * g1 + 8 + cs->w * g1;
    return ZKHIP_OK;
}
int32_t zkhip_setup_g16(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* toxic, const uint8_t* g1, const uint8_t* g2, uint8_t* pk_out,
                        uint64_t pk_cap) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(r1cs && toxic && pk_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "constraint system belongs to another context");
        ops_for(r1cs->curve)->setup(ctx, r1cs, toxic, g1, g2, pk_out, pk_cap);
    });
}

// ------------------------------------------------------------------ ZoKrates' own files (N1): host only, no context
int32_t zkhip_prog_parse(const uint8_t* bytes, size_t len, zkhip_prog** out) {
    if (!bytes || !out) { g_create_err = "null argument"; return ZKHIP_ERR_BAD_ARG; }
    *out = nullptr;
    return guarded_host([&] {
        std::unique_ptr<zkhip_prog> p(new zkhip_prog());
        prog_parse(bytes, len, p.get());
        *out = p.release();
    });
}
void zkhip_prog_free(zkhip_prog* prog) { delete prog; }
int32_t zkhip_prog_dims(const zkhip_prog* prog, uint64_t out[8]) {
    if (!prog || !out) return ZKHIP_ERR_BAD_ARG;
    out[0] = (uint64_t)prog->curve; out[1] = prog->n; out[2] = prog->l; out[3] = prog->w; out[4] = prog->return_count;
    out[5] = prog->public_args.size(); out[6] = prog->col[0].size() + prog->col[1].size() + prog->col[2].size(); out[7] = 0;
    return ZKHIP_OK;
}
int32_t zkhip_prog_matrix(const zkhip_prog* prog, int32_t which, const uint64_t** rowptr, const uint32_t** col, const uint8_t** val) {
    if (!prog || which < 0 || which > 2 || !rowptr || !col || !val) return ZKHIP_ERR_BAD_ARG;
    *rowptr = prog->rp[which].data();
    *col = prog->col[which].data();
    *val = prog->val[which].data();
    return ZKHIP_OK;
}
int32_t zkhip_prog_variable_order(const zkhip_prog* prog, const int64_t** ids) {
    if (!prog || !ids) return ZKHIP_ERR_BAD_ARG;
    *ids = prog->order.data();
    return ZKHIP_OK;
}
int32_t zkhip_prog_assignment(const zkhip_prog* prog, const uint8_t* witness, size_t len, uint8_t* z_out, uint8_t* inputs_out, uint64_t inputs_cap,
                              uint64_t* n_inputs) {
    if (!prog || !witness) { g_create_err = "null argument"; return ZKHIP_ERR_BAD_ARG; }
    return guarded_host([&] { prog_assignment(prog, witness, len, z_out, inputs_out, inputs_cap, n_inputs); });
}
int32_t zkhip_prog_write_bound(uint64_t n, uint64_t nnz, uint64_t n_args, uint64_t* bytes) {
    if (!bytes) return ZKHIP_ERR_BAD_ARG;
    *bytes = prog_write_bound(n, nnz, n_args);
    return ZKHIP_OK;
}
int32_t zkhip_prog_write(int32_t curve, uint64_t n, uint64_t m, const uint64_t* rowptr_a, const uint32_t* col_a, const uint8_t* val_a,
                         const uint64_t* rowptr_b, const uint32_t* col_b, const uint8_t* val_b, const uint64_t* rowptr_c, const uint32_t* col_c,
                         const uint8_t* val_c, const int64_t* ids, const int64_t* arg_ids, const uint8_t* arg_private, uint64_t n_args,
                         uint32_t return_count, uint8_t* out, uint64_t cap, uint64_t* len) {
    if (!rowptr_a || !rowptr_b || !rowptr_c || !ids || !out || !len || (n_args && (!arg_ids || !arg_private))) {
        g_create_err = "null argument";
        return ZKHIP_ERR_BAD_ARG;
    }
    return guarded_host([&] {
        const u64* rp[3] = {rowptr_a, rowptr_b, rowptr_c};
        const u32* col[3] = {col_a, col_b, col_c};
        const uint8_t* val[3] = {val_a, val_b, val_c};
        *len = prog_write(curve, n, m, rp, col, val, ids, arg_ids, arg_private, n_args, return_count, out, cap);
    });
}
int32_t zkhip_prog_r1cs_load(zkhip_ctx* ctx, const zkhip_prog* prog, zkhip_r1cs** out) {
    if (!ctx || !prog) return ZKHIP_ERR_BAD_ARG;
    return zkhip_r1cs_load(ctx, prog->curve, prog->n, prog->l, prog->w, prog->rp[0].data(), prog->col[0].data(), prog->val[0].data(),
                           prog->rp[1].data(), prog->col[1].data(), prog->val[1].data(), prog->rp[2].data(), prog->col[2].data(),
                           prog->val[2].data(), out);
}

// ------------------------------------------------------------------ N2: device-layout image of a loaded key
// "ZKHIPPK" + layout version; bump the version whenever the resident layout (packed points, sigma order, the extended base
// vectors, table levels) changes: an image is only meaningful to the library build that wrote it.
// An image holds level 0 of the five base tables; the window multiples are recomputed on the device at import (~0.1 s for a
// 2^20 key — less than reading the 6 GiB they occupy from any disk: measured in round 3, which is why the image that carried
// every level is gone).
static const char PK_IMAGE_MAGIC[8] = {'Z', 'K', 'H', 'I', 'P', 'P', 'K', '5'};
struct PkImageHeader {
    char magic[8];
    int32_t curve, scheme;
    uint64_t m, w, l, hlen, N;
    int32_t logN, c_z, c_h, sets;     // sets: s_z | s_h << 8 — which window multiples the tables hold (MsmShape::sets)
    uint32_t rank, world;
    int32_t ntt_split, reserved;      // the NTT split h_sigma is ordered for (zkhip_pk::ntt_log1 = NttPlan::split())
    uint64_t z_lo, z_n, h_lo, h_n;
    uint64_t len_delta, len_g2z2, len_buf[5];
    // version 5: level 0 of the bound tables H' / L' (this key's index ranges) when the key was bound at export, and the fingerprint
    // of the constraint system they were made for — zkhip_pk_bind_r1cs on the imported key attaches them when the system's
    // fingerprint agrees and costs a checksum instead of the transforms (0 / 0: the key was not bound)
    uint64_t len_bound[2];            // H', L'
    uint64_t bound_fp[2];
};
static DBuf* pk_bufs(zkhip_pk* pk, int k) { DBuf* b[5] = {&pk->a_ext, &pk->b1_ext, &pk->l_ext, &pk->b2_ext, &pk->h_sigma}; return b[k]; }
static int pk_levels(int curve, int c, int sets) {
    const int bits = ops_for(curve)->fr_bits, W = (bits + 1 + c - 1) / c;
    return (W + sets - 1) / sets;
}
// bytes of table k of this key: count x levels x point size (levels = 1: what an image holds)
static uint64_t pk_table_bytes(int curve, int k, uint64_t z_n, uint64_t h_n, int levels) {
    const uint64_t g1b = ops_for(curve)->packed_g1_bytes;
    const uint64_t count = std::max<uint64_t>(k == 4 ? h_n : z_n, 1), pt = k == 3 ? 2 * g1b : g1b;
    return count * pt * (uint64_t)levels;
}
int32_t zkhip_pk_export_size(const zkhip_pk* pk, uint64_t* bytes) {
    if (!pk || !bytes) return ZKHIP_ERR_BAD_ARG;
    uint64_t t = sizeof(PkImageHeader) + pk->delta_g1_canon.size() + pk->g_gamma2_z2_canon.size();
    for (int k = 0; k < 5; ++k) t += pk_table_bytes(pk->curve, k, pk->z_n, pk->h_n, 1);
    if (pk->h_bound.p && pk->l_bound.p) t += pk_table_bytes(pk->curve, 4, pk->z_n, pk->h_n, 1) + pk_table_bytes(pk->curve, 2, pk->z_n, pk->h_n, 1);
    *bytes = t;
    return ZKHIP_OK;
}
int32_t zkhip_pk_export(const zkhip_pk* pk_, uint8_t* out, uint64_t cap) {
    if (!pk_ || !out) return ZKHIP_ERR_BAD_ARG;
    zkhip_pk* pk = const_cast<zkhip_pk*>(pk_);
    zkhip_ctx* ctx = pk->ctx;
    return guarded(ctx, [&] {
        uint64_t need = 0;
        zkhip_pk_export_size(pk, &need);
        require(cap >= need, ZKHIP_ERR_BAD_ARG, "output buffer too small (see zkhip_pk_export_size)");
        PkImageHeader h;
        memset(&h, 0, sizeof(h));
        memcpy(h.magic, PK_IMAGE_MAGIC, 8);
        h.curve = pk->curve; h.scheme = pk->scheme;
        h.m = pk->m; h.w = pk->w; h.l = pk->l; h.hlen = pk->hlen; h.N = pk->N;
        h.logN = pk->logN; h.c_z = pk->c_z; h.c_h = pk->c_h; h.sets = pk->s_z | (pk->s_h << 8);
        h.rank = pk->rank; h.world = pk->world;
        h.ntt_split = pk->ntt_log1;
        h.z_lo = pk->z_lo; h.z_n = pk->z_n; h.h_lo = pk->h_lo; h.h_n = pk->h_n;
        h.len_delta = pk->delta_g1_canon.size(); h.len_g2z2 = pk->g_gamma2_z2_canon.size();
        for (int k = 0; k < 5; ++k) h.len_buf[k] = pk_table_bytes(pk->curve, k, pk->z_n, pk->h_n, 1);
        const bool with_bound = pk->h_bound.p && pk->l_bound.p;
        if (with_bound) {
            h.len_bound[0] = pk_table_bytes(pk->curve, 4, pk->z_n, pk->h_n, 1);
            h.len_bound[1] = pk_table_bytes(pk->curve, 2, pk->z_n, pk->h_n, 1);
            h.bound_fp[0] = pk->bound_fp[0]; h.bound_fp[1] = pk->bound_fp[1];
        }
        uint8_t* p = out;
        memcpy(p, &h, sizeof(h)); p += sizeof(h);
        memcpy(p, pk->delta_g1_canon.data(), h.len_delta); p += h.len_delta;
        memcpy(p, pk->g_gamma2_z2_canon.data(), h.len_g2z2); p += h.len_g2z2;
        for (int k = 0; k < 5; ++k) {
            dev_d2h(p, pk_bufs(pk, k)->p, h.len_buf[k], ctx->stream);     // level 0 leads every table
            p += h.len_buf[k];
        }
        if (with_bound) {
            dev_d2h(p, pk->h_bound.p, h.len_bound[0], ctx->stream); p += h.len_bound[0];
            dev_d2h(p, pk->l_bound.p, h.len_bound[1], ctx->stream); p += h.len_bound[1];
        }
        stream_sync(ctx->stream);
    });
}
int32_t zkhip_pk_import(zkhip_ctx* ctx, const uint8_t* bytes, size_t len, zkhip_pk** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(bytes && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        require(len >= sizeof(PkImageHeader), ZKHIP_ERR_PARSE, "key image truncated");
        PkImageHeader h;
        memcpy(&h, bytes, sizeof(h));
        require(!memcmp(h.magic, PK_IMAGE_MAGIC, 8), ZKHIP_ERR_PARSE, "not a key image of this library version (re-import the proving key)");
        const CurveOps* ops = ops_for(h.curve);   // (validates the curve id)
        require(h.scheme == 0 || h.scheme == 1, ZKHIP_ERR_PARSE, "key image: unknown scheme");
        uint64_t total = sizeof(PkImageHeader), rest = len - sizeof(PkImageHeader);
        const uint64_t parts[9] = {h.len_delta, h.len_g2z2, h.len_buf[0], h.len_buf[1], h.len_buf[2], h.len_buf[3], h.len_buf[4], h.len_bound[0], h.len_bound[1]};
        for (uint64_t part : parts) {
            require(part <= rest, ZKHIP_ERR_PARSE, "key image truncated");
            rest -= part;
            total += part;
        }
        require(total == len, ZKHIP_ERR_PARSE, "trailing bytes after key image");
        int s_z = h.sets & 0xff, s_h = (h.sets >> 8) & 0xff;
        require(h.world >= 1 && h.rank < h.world && h.logN >= 0 && h.logN <= 3 * NTT_MAX_SUBLOG && h.N == ((uint64_t)1 << h.logN) && h.z_n <= h.m + 2 &&
                    h.h_n <= h.N && h.c_z >= 2 && h.c_z <= MSM_MAX_C && h.c_h >= 2 && h.c_h <= MSM_MAX_C && s_z >= 1 && s_h >= 1 && (h.sets >> 16) == 0,
                ZKHIP_ERR_PARSE, "key image: inconsistent header");
        // the index ranges must lie inside the key and the five base arrays must have exactly the size the ranges imply:
        // the kernels trust these numbers
        bool sizes_ok = h.m + 2 < ((uint64_t)1 << 31) && h.z_lo <= h.m + 2 && h.z_n <= h.m + 2 - h.z_lo && h.h_lo <= h.N && h.h_n <= h.N - h.h_lo &&
                        h.len_delta <= 4096 && h.len_g2z2 <= 4096;
        for (int k = 0; k < 5 && sizes_ok; ++k) sizes_ok = h.len_buf[k] == pk_table_bytes(h.curve, k, h.z_n, h.h_n, 1);
        const bool with_bound = h.len_bound[0] || h.len_bound[1];
        if (with_bound)
            sizes_ok = sizes_ok && h.len_bound[0] == pk_table_bytes(h.curve, 4, h.z_n, h.h_n, 1) && h.len_bound[1] == pk_table_bytes(h.curve, 2, h.z_n, h.h_n, 1) &&
                       (h.bound_fp[0] | h.bound_fp[1]) != 0;
        require(sizes_ok, ZKHIP_ERR_PARSE, "key image: array sizes do not match the header");
        require(ops->ntt_log1(ctx, h.logN) == h.ntt_split, ZKHIP_ERR_PARSE,
                "key image: written under another NTT split (NTT_SINGLE_MAX_LOG / NTT_MAX_SUBLOG) than this context uses; re-import the proving key");
        // The image carries level 0 only and the window multiples are recomputed here, so how many of them THIS device keeps is this
        // context's decision, not the exporter's: ZKHIP_TUNE_MSM_SETS if set, else the header's count, doubled until the tables fit
        // 60 % of the free device memory (PkLoader::finish_tables' rule: an image written on an empty 288 GB device must still load
        // beside other tenants, with more bucket sets instead of an allocation failure).
        {
            const int W_z = (ops->fr_bits + 1 + h.c_z - 1) / h.c_z, W_h = (ops->fr_bits + 1 + h.c_h - 1) / h.c_h;
            if (ctx->msm_sets) s_z = std::min(ctx->msm_sets, W_z), s_h = std::min(ctx->msm_sets, W_h);
            const uint64_t budget = msm_table_budget(ctx, h.z_n, h.h_n, h.N, W_z, 1u << (h.c_z - 1));
            while (!ctx->msm_sets) {
                uint64_t need = 0;
                for (int k = 0; k < 5; ++k) need += pk_table_bytes(h.curve, k, h.z_n, h.h_n, k == 4 ? pk_levels(h.curve, h.c_h, s_h) : pk_levels(h.curve, h.c_z, s_z));
                if (need <= budget || (s_z >= W_z && s_h >= W_h)) break;
                s_z = std::min(2 * s_z, W_z);
                s_h = std::min(2 * s_h, W_h);
            }
        }
        // (c, sets) must be a shape this context's sort can run
        require(ops->msm_shape_ok(ctx, h.z_n, h.c_z, s_z) && ops->msm_shape_ok(ctx, h.h_n, h.c_h, s_h), ZKHIP_ERR_PARSE,
                "key image: window width / bucket sets not usable under this context's settings; re-import the proving key");
        std::unique_ptr<zkhip_pk> pk(new zkhip_pk());
        pk->curve = h.curve; pk->scheme = h.scheme; pk->ctx = ctx;
        pk->m = h.m; pk->w = h.w; pk->l = h.l; pk->hlen = h.hlen; pk->N = h.N; pk->logN = h.logN;
        pk->c_z = h.c_z; pk->c_h = h.c_h; pk->s_z = s_z; pk->s_h = s_h; pk->rank = h.rank; pk->world = h.world;
        pk->z_lo = h.z_lo; pk->z_n = h.z_n; pk->h_lo = h.h_lo; pk->h_n = h.h_n;
        pk->ntt_log1 = h.ntt_split;
        const uint8_t* p = bytes + sizeof(h);
        pk->delta_g1_canon.assign(p, p + h.len_delta); p += h.len_delta;
        pk->g_gamma2_z2_canon.assign(p, p + h.len_g2z2); p += h.len_g2z2;
        for (int k = 0; k < 5; ++k) {
            DBuf* b = pk_bufs(pk.get(), k);
            b->ensure(pk_table_bytes(h.curve, k, h.z_n, h.h_n, k == 4 ? pk_levels(h.curve, h.c_h, s_h) : pk_levels(h.curve, h.c_z, s_z)));
            const uint8_t* src = p;
            dev_h2d_fill(b->p, h.len_buf[k], 64, ctx->stream, [src](char* out, size_t off, size_t len) { memcpy(out, src + off, len); });
            p += h.len_buf[k];
        }
        stream_sync(ctx->stream);
        ops->pk_table_levels(ctx, pk.get());       // recompute the window multiples behind level 0
        if (with_bound) {                          // ... and behind level 0 of H' / L'; attached to a system by zkhip_pk_bind_r1cs (fingerprint)
            ops->install_bound(ctx, pk.get(), p, p + h.len_bound[0], false, h.bound_fp);
            p += h.len_bound[0] + h.len_bound[1];
        }
        *out = pk.release();
    });
}

// ------------------------------------------------------------------ one proof across several GPUs of one process
struct zkhip_multi {
    std::vector<zkhip_ctx*> ctx;     // one per member; a device may appear more than once
    std::vector<zkhip_pk*> pk;       // member k: shard k of n
    std::vector<zkhip_r1cs*> cs;     // replicas
    int scheme = -1;                 // of the loaded key: 0 Groth16, 1 GM17
    bool replicas = false;           // every member holds the WHOLE key (throughput mode) instead of shard k of n
    bool transform_split = true;     // bound members split the witness map (even members transform a, odd ones b, partners exchange)
    std::string err;
    // the exchange step over RCCL (zkhip_multi_use_rccl): one communicator and one pair of device buffers per member
    bool rccl = false;
    std::vector<void*> comm;         // ncclComm_t per member
    std::vector<DBuf> gather1, gather2;   // all members' ws1 / ws2, rank-major, on every member's device
    std::string rccl_desc;
    bool last_split = false;         // the last zkhip_prove_*_multi split its witness map
};
}  // extern "C"

// ------------------------------------------------------------------ RCCL, bound at run time
// The data path of a sharded proof has ONE exchange: every member contributes the bucket-set sums of its five partial
// MSMs ("ncclAllGather of a fixed-size struct").  libzkhip does not link librccl — a single-GPU prover, the
// CLI, the tests never need it and would pay its load time — so the five entry points used are resolved with dlopen when a
// caller asks for the RCCL exchange.  Whatever HIP runtime this process already runs is the one librccl.so.1 binds to.
#ifndef ZK_EMU
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string why;                 // why it is unavailable
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) { x.why = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?"); return x; }
        auto sym = [&](const char* n) { void* p = dlsym(x.handle, n); if (!p && x.why.empty()) x.why = std::string("librccl lacks ") + n; return p; };
        x.CommInitAll = (decltype(x.CommInitAll))sym("ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
        x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
        x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
        x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
        x.GetVersion = (decltype(x.GetVersion))sym("ncclGetVersion");
        return x;
    }();
    return r;
}
void rccl_check(ncclResult_t rc, const char* what) {
    if (rc != ncclSuccess) throw DevError{std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error")};
}
}  // namespace
#endif
// run fn(k) for every member — one host thread each (contexts are independent; the test emulator is single-threaded) —
// and keep the first failure
template <class Fn>
static int32_t multi_each(zkhip_multi* m, Fn&& fn) try {
    const size_t n = m->ctx.size();
    std::vector<int32_t> rc(n, ZKHIP_OK);
    std::vector<std::string> msg(n);
    auto body = [&](size_t k) {
        rc[k] = fn(k);
        if (rc[k] != ZKHIP_OK) msg[k] = zkhip_last_error(m->ctx[k]);
    };
#ifdef ZK_EMU
    for (size_t k = 0; k < n; ++k) body(k);
#else
    {
        HostThreads th;
        for (size_t k = 1; k < n; ++k) th.run([&, k] { body(k); });
        body(0);
    }
#endif
    for (size_t k = 0; k < n; ++k)
        if (rc[k] != ZKHIP_OK) {
            m->err = "member " + std::to_string(k) + ": " + msg[k];
            return rc[k];
        }
    return ZKHIP_OK;
} catch (const std::bad_alloc&) {   // nothing may escape through the C ABI (HostThreads joins what was started)
    return ZKHIP_ERR_NOMEM;
} catch (...) {
    return ZKHIP_ERR_DEVICE;
}
// every member holds a constraint system and a key
static bool multi_loaded(const zkhip_multi* m) {
    for (auto* c : m->cs) if (!c) return false;
    for (auto* p : m->pk) if (!p) return false;
    return true;
}
extern "C" {
static void multi_drop_keys(zkhip_multi* m) {
    for (auto*& p : m->pk) { zkhip_pk_free(p); p = nullptr; }
    m->scheme = -1;
    m->replicas = false;
}
int32_t zkhip_ctx_create_multi(const int32_t* devices, int32_t n, zkhip_multi** out) {
    if (!out) { g_create_err = "out is NULL"; return ZKHIP_ERR_BAD_ARG; }
    *out = nullptr;
    if (!devices || n < 1 || n > 64) { g_create_err = "device list empty or longer than 64"; return ZKHIP_ERR_BAD_ARG; }
    std::unique_ptr<zkhip_multi> m(new (std::nothrow) zkhip_multi());
    if (!m) { g_create_err = "out of host memory"; return ZKHIP_ERR_NOMEM; }
    try {
        m->ctx.reserve(n);
        m->pk.assign(n, nullptr);
        m->cs.assign(n, nullptr);
    } catch (const std::bad_alloc&) {
        g_create_err = "out of host memory";
        return ZKHIP_ERR_NOMEM;
    }
    for (int32_t k = 0; k < n; ++k) {
        zkhip_ctx* c = nullptr;
        const int32_t rc = zkhip_ctx_create(devices[k], &c);
        if (rc != ZKHIP_OK) {
            for (auto* q : m->ctx) zkhip_ctx_free(q);
            return rc;        // message already in the per-thread create error
        }
        m->ctx.push_back(c);  // (capacity reserved above)
    }
    // members that share a device share its memory: each sizes its tables for its share of what is free (msm_table_budget)
    for (int32_t k = 0; k < n; ++k) {
        int same = 0;
        for (int32_t q = 0; q < n; ++q) same += devices[q] == devices[k];
        m->ctx[k]->tenants = same;
    }
    *out = m.release();
    return ZKHIP_OK;
}
static void multi_drop_rccl(zkhip_multi* m) {
#ifndef ZK_EMU
    for (size_t k = 0; k < m->comm.size(); ++k)
        if (m->comm[k]) {
            try { dev_set(m->ctx[k]->device); } catch (...) {}
            (void)rccl().CommDestroy((ncclComm_t)m->comm[k]);
        }
#endif
    m->comm.clear();
    for (size_t k = 0; k < m->gather1.size(); ++k) {
        try { dev_set(m->ctx[k]->device); } catch (...) {}
        m->gather1[k].release();
        m->gather2[k].release();
    }
    m->gather1.clear();
    m->gather2.clear();
    m->rccl = false;
}
int32_t zkhip_multi_use_rccl(zkhip_multi* m, int32_t on) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    try {
        multi_drop_rccl(m);
        if (!on) return ZKHIP_OK;
        const size_t n = m->ctx.size();
        std::vector<int> dev(n);
        for (size_t k = 0; k < n; ++k) dev[k] = m->ctx[k]->device;
#ifndef ZK_EMU   // (the emulator's "all-gather" is a loop of copies: its members may share the one emulated device)
        for (size_t a = 0; a < n; ++a)
            for (size_t b = a + 1; b < n; ++b)
                if (dev[a] == dev[b]) {
                    m->err = "RCCL wants one device per member (device " + std::to_string(dev[a]) + " is listed twice): the host exchange stays in use";
                    return ZKHIP_ERR_BAD_ARG;
                }
#endif
        m->gather1 = std::vector<DBuf>(n);
        m->gather2 = std::vector<DBuf>(n);
#ifdef ZK_EMU
        m->rccl_desc = "emulated all-gather (TEST EMULATOR: device-to-device copies in member order)";
#else
        Rccl& R = rccl();
        if (!R.why.empty()) { m->err = R.why; m->gather1.clear(); m->gather2.clear(); return ZKHIP_ERR_DEVICE; }
        std::vector<ncclComm_t> comms(n, nullptr);
        rccl_check(R.CommInitAll(comms.data(), (int)n, dev.data()), "ncclCommInitAll");
        m->comm.assign(comms.begin(), comms.end());
        int ver = 0;
        (void)R.GetVersion(&ver);
        m->rccl_desc = "RCCL " + std::to_string(ver) + ", " + std::to_string(n) + " rank(s), ncclAllGather of the members' bucket-set sums";
        for (size_t a = 0; a < n; ++a)          // xGMI peer reachability, for the record (RCCL picks its own transport)
            for (size_t b = 0; b < n; ++b) {
                int can = 0;
                if (a != b && hipDeviceCanAccessPeer(&can, dev[a], dev[b]) == hipSuccess && !can) m->rccl_desc += "; no peer access " + std::to_string(dev[a]) + "->" + std::to_string(dev[b]);
            }
#endif
        m->rccl = true;
        return ZKHIP_OK;
    } catch (const DevError& e) {
        m->err = e.msg;
        multi_drop_rccl(m);
        return ZKHIP_ERR_DEVICE;
    } catch (const std::bad_alloc&) {
        m->err = "out of host memory";
        multi_drop_rccl(m);
        return ZKHIP_ERR_NOMEM;
    } catch (...) {
        m->err = "unexpected internal error";
        multi_drop_rccl(m);
        return ZKHIP_ERR_DEVICE;
    }
}
const char* zkhip_multi_exchange(const zkhip_multi* m) {
    if (!m) return "";
    return m->rccl ? m->rccl_desc.c_str() : "canonical records in host memory, combined on the calling thread";
}
void zkhip_multi_free(zkhip_multi* m) {
    if (!m) return;
    multi_drop_rccl(m);
    multi_drop_keys(m);
    for (auto* c : m->cs) zkhip_r1cs_free(c);
    for (auto* c : m->ctx) zkhip_ctx_free(c);
    delete m;
}
int32_t zkhip_multi_size(const zkhip_multi* m) { return m ? (int32_t)m->ctx.size() : 0; }
zkhip_ctx* zkhip_multi_ctx(zkhip_multi* m, int32_t member) { return (m && member >= 0 && (size_t)member < m->ctx.size()) ? m->ctx[member] : nullptr; }
const char* zkhip_multi_last_error(const zkhip_multi* m) { return m ? m->err.c_str() : g_create_err.c_str(); }
int32_t zkhip_multi_r1cs_load(zkhip_multi* m, int32_t curve, uint64_t n, uint64_t l, uint64_t w, const uint64_t* rowptr_a, const uint32_t* col_a,
                              const uint8_t* val_a, const uint64_t* rowptr_b, const uint32_t* col_b, const uint8_t* val_b, const uint64_t* rowptr_c,
                              const uint32_t* col_c, const uint8_t* val_c) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    auto drop = [&] { for (auto*& c : m->cs) { zkhip_r1cs_free(c); c = nullptr; } };
    drop();
    const int32_t rc = multi_each(m, [&](size_t k) {
        return zkhip_r1cs_load(m->ctx[k], curve, n, l, w, rowptr_a, col_a, val_a, rowptr_b, col_b, val_b, rowptr_c, col_c, val_c, &m->cs[k]);
    });
    if (rc != ZKHIP_OK) drop();      // all members or none: a partly loaded group must not reach the provers
    return rc;
}
int32_t zkhip_multi_pk_load_g16(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    multi_drop_keys(m);
    const uint32_t world = (uint32_t)m->ctx.size();
    const int32_t rc = multi_each(m, [&](size_t k) { return zkhip_pk_load_g16_shard(m->ctx[k], curve, bytes, len, (uint32_t)k, world, &m->pk[k]); });
    if (rc == ZKHIP_OK) m->scheme = 0; else multi_drop_keys(m);
    return rc;
}
int32_t zkhip_multi_pk_load_gm17(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    multi_drop_keys(m);
    const uint32_t world = (uint32_t)m->ctx.size();
    const int32_t rc = multi_each(m, [&](size_t k) { return zkhip_pk_load_gm17_shard(m->ctx[k], curve, bytes, len, (uint32_t)k, world, &m->pk[k]); });
    if (rc == ZKHIP_OK) m->scheme = 1; else multi_drop_keys(m);
    return rc;
}
// The members' keys bound to the members' constraint system: member 0 computes level 0 of H' / L' over the whole index range from
// the key file (the same bytes the key was loaded from), every member installs its own ranges (replicas: all of it).
int32_t zkhip_multi_bind(zkhip_multi* m, const uint8_t* key_bytes, size_t len) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    if (!key_bytes) { m->err = "null argument"; return ZKHIP_ERR_BAD_ARG; }
    if (m->scheme < 0 || !multi_loaded(m)) { m->err = "load the constraint system and a proving key first"; return ZKHIP_ERR_BAD_ARG; }
    std::vector<uint8_t> h_host, l_host;
    u64 fp[2] = {0, 0};
    const CurveOps* ops = nullptr;
    int32_t rc = guarded(m->ctx[0], [&] {
        ops = ops_for(m->pk[0]->curve);
        for (size_t k = 0; k < m->ctx.size(); ++k) ops->pk_bind_check(m->ctx[k], m->pk[k], m->cs[k]);
        ops->bound_level0_from_file(m->ctx[0], m->scheme, m->cs[0], key_bytes, len, h_host, l_host, fp);
    });
    if (rc != ZKHIP_OK) { m->err = std::string("member 0: ") + zkhip_last_error(m->ctx[0]); return rc; }
    rc = multi_each(m, [&](size_t k) {
        return guarded(m->ctx[k], [&] { ops->install_bound_ranges(m->ctx[k], m->pk[k], m->cs[k], h_host.data(), h_host.size(), l_host.data(), l_host.size(), fp); });
    });
    if (rc != ZKHIP_OK) for (auto* p : m->pk) zkhip_pk_unbind(p);      // all members or none
    return rc;
}
// 1 / 0: let bound members split the witness map of a proof between them (default 1); returns what it was.  -1: only report.
int32_t zkhip_multi_transform_split(zkhip_multi* m, int32_t on) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    const int32_t was = m->transform_split ? 1 : 0;
    if (on >= 0) m->transform_split = on != 0;
    return was;
}
// 1 if the last zkhip_prove_*_multi split its witness map between the members, else 0
int32_t zkhip_multi_last_split(const zkhip_multi* m) { return m && m->last_split ? 1 : 0; }
int32_t zkhip_multi_unbind(zkhip_multi* m) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    for (auto* p : m->pk) if (p) zkhip_pk_unbind(p);
    return ZKHIP_OK;
}
// throughput mode: the whole key on every member; zkhip_prove_g16_multi_batch deals independent proofs round-robin
int32_t zkhip_multi_pk_load_g16_replicas(zkhip_multi* m, int32_t curve, const uint8_t* bytes, size_t len) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    multi_drop_keys(m);
    const int32_t rc = multi_each(m, [&](size_t k) { return zkhip_pk_load_g16(m->ctx[k], curve, bytes, len, &m->pk[k]); });
    if (rc == ZKHIP_OK) { m->scheme = 0; m->replicas = true; } else multi_drop_keys(m);
    return rc;
}
int32_t zkhip_prove_g16_multi_batch(zkhip_multi* m, uint32_t count, const uint8_t* z, const uint8_t* rs, uint8_t* proofs_out, zkhip_timings* timings) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    if ((!z || !rs || !proofs_out) && count) { m->err = "null argument"; return ZKHIP_ERR_BAD_ARG; }
    if (m->scheme != 0 || !m->replicas || !multi_loaded(m)) {
        m->err = "load the constraint system and zkhip_multi_pk_load_g16_replicas first";
        return ZKHIP_ERR_BAD_ARG;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n = m->ctx.size();
    const uint64_t zb = (uint64_t)m->pk[0]->m * 32;
    const size_t proof_bytes = (m->pk[0]->curve == ZKHIP_CURVE_BN128 ? 32 : 48) * 8 + 3;
    // member k proves the contiguous block [lo_k, hi_k) of the batch through its own pipelined batch call
    std::vector<zkhip_timings> tm(n);
    const int32_t rc = multi_each(m, [&](size_t k) {
        const uint64_t lo = (uint64_t)count * k / n, hi = (uint64_t)count * (k + 1) / n;
        memset(&tm[k], 0, sizeof(tm[k]));
        if (hi == lo) return (int32_t)ZKHIP_OK;
        return zkhip_prove_g16_batch(m->ctx[k], m->pk[k], m->cs[k], (uint32_t)(hi - lo), z + lo * zb, rs + lo * 64, proofs_out + lo * proof_bytes, &tm[k]);
    });
    if (rc != ZKHIP_OK) return rc;
    if (timings) {
        memset(timings, 0, sizeof(*timings));
        for (size_t k = 0; k < n; ++k) {
            float* a = (float*)timings; const float* b = (const float*)&tm[k];
            for (size_t q = 0; q < sizeof(zkhip_timings) / sizeof(float); ++q) a[q] += b[q];
        }
        timings->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return ZKHIP_OK;
}
static int32_t multi_prove(zkhip_multi* m, int scheme, const uint8_t* z, const uint8_t* rnd, const uint8_t* s_, uint8_t* proof_out, zkhip_timings* timings) {
    if (!m) return ZKHIP_ERR_BAD_ARG;
    if (!z || !rnd || !proof_out || (scheme == 0 && !s_)) { m->err = "null argument"; return ZKHIP_ERR_BAD_ARG; }
    if (m->scheme != scheme || !multi_loaded(m)) { m->err = "load the constraint system and a proving key of this scheme first"; return ZKHIP_ERR_BAD_ARG; }
    if (m->replicas) { m->err = "the members hold whole keys (replicas): use zkhip_prove_g16_multi_batch, or load the key sharded"; return ZKHIP_ERR_BAD_ARG; }
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t rec = 0;
    zkhip_partial_size(m->pk[0]->curve, &rec);
    const size_t n = m->ctx.size();
    std::vector<uint8_t> records(n * rec);
    std::vector<zkhip_timings> tm(n);
    int32_t rc;
    // The witness map SPLIT between the members (SURVEY.md §8e / north_star "NTT domain shard"): over keys bound to the system a proof
    // needs a and b on the coset and nothing else — members of even rank transform a, those of odd rank b (two transforms instead
    // of four each), partners copy each other's vector (N x 32 B over xGMI, or on the device they share) and every member multiplies.
    // Unbound keys, GM17, one member, domains below 2^18: every member runs the whole map, as before.
    const CurveOps* sops = ops_for(m->pk[0]->curve);
    bool split = scheme == 0 && n >= 2 && m->transform_split;
    for (size_t k = 0; k < n && split; ++k) split = sops->can_split(m->ctx[k], m->pk[k], m->cs[k]);
    auto half_of = [](size_t k) { return (int)(k & 1); };
    auto partner_of = [n](size_t k) { return (k ^ 1) < n ? (k ^ 1) : k - 1; };
    auto fetch = [&](size_t k) {      // (inside guarded(m->ctx[k], ...))
        const size_t p = partner_of(k);
        sops->split_fetch(m->ctx[k], m->pk[k], sops->split_half_ptr(m->ctx[p], m->pk[p], half_of(p)), m->ctx[p]->device, m->ctx[p]->slots[0].half_ready);
    };
    m->last_split = split;
    if (split) {
        // phase A on every member before phase B on any: a member fetches its partner's half only when that half is enqueued
        rc = multi_each(m, [&](size_t k) {
            return guarded(m->ctx[k], [&] {
                require(m->pk[k]->ctx == m->ctx[k] && m->cs[k]->ctx == m->ctx[k], ZKHIP_ERR_BAD_ARG, "handles belong to another context");
                for (auto& sl : m->ctx[k]->slots) require(!sl.busy, ZKHIP_ERR_BAD_ARG, "a proof is in flight in this context");
                sops->split_begin(m->ctx[k], m->pk[k], m->cs[k], z, nullptr, rnd, s_, half_of(k));
            });
        });
        if (rc != ZKHIP_OK) { for (auto* c : m->ctx) { try { dev_set(c->device); dev_sync_all(); } catch (...) {} } return rc; }
    }
    if (m->rccl) {
        // every member leaves the bucket-set sums of its five partial MSMs on its device and all-gathers them (RCCL over
        // xGMI; in the test emulator: copies in member order); member 0's gathered copy is read back and combined
        const CurveOps* ops = ops_for(m->pk[0]->curve);
        std::vector<const void*> d1(n), d2(n);
        std::vector<size_t> b1(n), b2(n);
        auto share = [&](size_t k) {
            return guarded(m->ctx[k], [&] {
                zkhip_ctx* c = m->ctx[k];
                require(m->pk[k]->ctx == c && m->cs[k]->ctx == c, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
                if (split) { fetch(k); ops->split_end_device_sums(c, m->pk[k], m->cs[k], &d1[k], &b1[k], &d2[k], &b2[k], &tm[k]); }
                else if (scheme == 0) ops->prove_device_sums(c, m->pk[k], m->cs[k], z, rnd, s_, &d1[k], &b1[k], &d2[k], &b2[k], &tm[k]);
                else ops->gm17_prove_device_sums(c, m->pk[k], m->cs[k], z, rnd, &d1[k], &b1[k], &d2[k], &b2[k], &tm[k]);
                m->gather1[k].ensure(n * b1[k]);
                m->gather2[k].ensure(n * b2[k]);
            });
        };
#ifdef ZK_EMU
        rc = multi_each(m, share);
        if (rc != ZKHIP_OK) return rc;
        for (size_t k = 0; k < n; ++k) {
            dev_d2d((uint8_t*)m->gather1[0].p + k * b1[0], d1[k], b1[0], m->ctx[0]->stream);
            dev_d2d((uint8_t*)m->gather2[0].p + k * b2[0], d2[k], b2[0], m->ctx[0]->stream);
        }
#else
        // two phases: a member enters the collective only when EVERY member has its share (one that failed alone would
        // leave the others waiting in ncclAllGather for ever)
        rc = multi_each(m, share);
        if (rc != ZKHIP_OK) return rc;
        rc = multi_each(m, [&](size_t k) {
            return guarded(m->ctx[k], [&] {
                Rccl& R = rccl();
                zkhip_ctx* c = m->ctx[k];
                rccl_check(R.GroupStart(), "ncclGroupStart");
                rccl_check(R.AllGather(d1[k], m->gather1[k].p, b1[k], ncclUint8, (ncclComm_t)m->comm[k], c->stream), "ncclAllGather");
                rccl_check(R.AllGather(d2[k], m->gather2[k].p, b2[k], ncclUint8, (ncclComm_t)m->comm[k], c->stream), "ncclAllGather");
                rccl_check(R.GroupEnd(), "ncclGroupEnd");
                stream_sync(c->stream);
            });
        });
        if (rc != ZKHIP_OK) return rc;
#endif
        rc = guarded(m->ctx[0], [&] {
            std::vector<uint8_t> h1(n * b1[0]), h2(n * b2[0]);
            dev_d2h(h1.data(), m->gather1[0].p, h1.size(), m->ctx[0]->stream);
            dev_d2h(h2.data(), m->gather2[0].p, h2.size(), m->ctx[0]->stream);
            stream_sync(m->ctx[0]->stream);
            for (size_t k = 0; k < n; ++k) ops->record_from_sums(m->ctx[k], m->pk[k], &h1[k * b1[0]], &h2[k * b2[0]], &records[k * rec]);
        });
        if (rc != ZKHIP_OK) { m->err = std::string("gather: ") + zkhip_last_error(m->ctx[0]); return rc; }
    } else {
        rc = multi_each(m, [&](size_t k) {
            if (split) return guarded(m->ctx[k], [&] { fetch(k); sops->split_end_partial(m->ctx[k], m->pk[k], m->cs[k], &records[k * rec], &tm[k]); });
            return scheme == 0 ? zkhip_prove_g16_partial(m->ctx[k], m->pk[k], m->cs[k], z, nullptr, rnd, s_, &records[k * rec], &tm[k])
                               : zkhip_prove_gm17_partial(m->ctx[k], m->pk[k], m->cs[k], z, nullptr, rnd, &records[k * rec], &tm[k]);
        });
        if (rc != ZKHIP_OK) return rc;
    }
    rc = scheme == 0 ? zkhip_combine_g16(m->ctx[0], m->pk[0], (uint32_t)n, records.data(), rnd, s_, proof_out)
                     : zkhip_combine_gm17(m->ctx[0], m->pk[0], (uint32_t)n, records.data(), rnd, proof_out);
    if (rc != ZKHIP_OK) { m->err = std::string("combine: ") + zkhip_last_error(m->ctx[0]); return rc; }
    if (timings) {   // the slowest member per phase; total = wall clock of the whole call
        *timings = tm[0];
        for (size_t k = 1; k < n; ++k) {
            float* a = (float*)timings; const float* b = (const float*)&tm[k];
            for (size_t q = 0; q < sizeof(zkhip_timings) / sizeof(float); ++q) a[q] = std::max(a[q], b[q]);
        }
        timings->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return ZKHIP_OK;
}
int32_t zkhip_prove_g16_multi(zkhip_multi* m, const uint8_t* z, const uint8_t* r, const uint8_t* s, uint8_t* proof_out, zkhip_timings* timings) {
    return multi_prove(m, 0, z, r, s, proof_out, timings);
}
int32_t zkhip_prove_gm17_multi(zkhip_multi* m, const uint8_t* z, const uint8_t* d1_d2_r, uint8_t* proof_out, zkhip_timings* timings) {
    return multi_prove(m, 1, z, d1_d2_r, nullptr, proof_out, timings);
}

// ------------------------------------------------------------------ GM17 (config 5)
int32_t zkhip_pk_load_gm17(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, zkhip_pk** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(bytes && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        std::unique_ptr<zkhip_pk> pk(new zkhip_pk());
        pk->curve = curve;
        pk->ctx = ctx;
        ops_for(curve)->gm17_pk_load(ctx, bytes, len, pk.get());
        *out = pk.release();
    });
}
int32_t zkhip_pk_load_gm17_shard(zkhip_ctx* ctx, int32_t curve, const uint8_t* bytes, size_t len, uint32_t rank, uint32_t world, zkhip_pk** out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(bytes && out, ZKHIP_ERR_BAD_ARG, "null argument");
        *out = nullptr;
        require(world >= 1 && world <= 64 && rank < world, ZKHIP_ERR_BAD_ARG, "rank / world out of range (1 <= world <= 64)");
        std::unique_ptr<zkhip_pk> pk(new zkhip_pk());
        pk->curve = curve;
        pk->ctx = ctx;
        pk->rank = rank;
        pk->world = world;
        ops_for(curve)->gm17_pk_load(ctx, bytes, len, pk.get());
        *out = pk.release();
    });
}
int32_t zkhip_prove_gm17_partial(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, zkhip_assignment* z_resident,
                                 const uint8_t* d1_d2_r, uint8_t* partial_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && (z || z_resident) && d1_d2_r && partial_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        if (!z) require(z_resident->ctx == ctx && z_resident->curve == pk->curve && z_resident->m == r1cs->l + r1cs->w, ZKHIP_ERR_BAD_ARG,
                        "assignment does not match the constraint system");
        ops_for(pk->curve)->gm17_prove_partial(ctx, pk, r1cs, z, z ? nullptr : z_resident->scalars.p, d1_d2_r, partial_out, timings);
    });
}
int32_t zkhip_combine_gm17(zkhip_ctx* ctx, const zkhip_pk* pk, uint32_t count, const uint8_t* partials, const uint8_t* d1_d2_r, uint8_t* proof_out) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && partials && d1_d2_r && proof_out && count >= 1, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->scheme == 1, ZKHIP_ERR_BAD_ARG, "this is a Groth16 proving key");
        ops_for(pk->curve)->gm17_combine(pk, count, partials, d1_d2_r, proof_out);
    });
}
int32_t zkhip_prove_gm17(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, const uint8_t* z, const uint8_t* d1_d2_r,
                         uint8_t* proof_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && z && d1_d2_r && proof_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        ops_for(pk->curve)->gm17_prove(ctx, pk, r1cs, z, nullptr, d1_d2_r, proof_out, timings);
    });
}
int32_t zkhip_prove_gm17_resident(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, zkhip_assignment* z, const uint8_t* d1_d2_r,
                                  uint8_t* proof_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && z && d1_d2_r && proof_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx && z->ctx == ctx, ZKHIP_ERR_BAD_ARG, "handles belong to another context");
        require(z->curve == pk->curve && z->m == r1cs->l + r1cs->w, ZKHIP_ERR_BAD_ARG, "assignment does not match the constraint system");
        ops_for(pk->curve)->gm17_prove(ctx, pk, r1cs, nullptr, z->scalars.p, d1_d2_r, proof_out, timings);
    });
}
int32_t zkhip_prove_gm17_resident_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* r1cs, uint32_t count, zkhip_assignment* const* zs,
                                        const uint8_t* d1_d2_r, uint8_t* proofs_out, zkhip_timings* timings) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(pk && r1cs && d1_d2_r && proofs_out && (zs || count == 0), ZKHIP_ERR_BAD_ARG, "null argument");
        require(pk->ctx == ctx && r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "key / constraint system belong to another context");
        std::vector<void*> dev(count);
        for (uint32_t i = 0; i < count; ++i) {
            require(zs[i] && zs[i]->ctx == ctx && zs[i]->curve == pk->curve && zs[i]->m == r1cs->l + r1cs->w, ZKHIP_ERR_BAD_ARG,
                    "assignment does not match the constraint system");
            dev[i] = zs[i]->scalars.p;
        }
        ops_for(pk->curve)->gm17_prove_batch(ctx, pk, r1cs, count, nullptr, dev.data(), d1_d2_r, proofs_out, timings);
    });
}
int32_t zkhip_setup_gm17_size(const zkhip_r1cs* cs, uint64_t* pk_bytes) {
    if (!cs || !pk_bytes) return ZKHIP_ERR_BAD_ARG;
    try {
        *pk_bytes = ops_for(cs->curve)->gm17_key_bytes(cs->n, cs->l, cs->w);
    } catch (...) {
        return ZKHIP_ERR_BAD_ARG;
    }
    return ZKHIP_OK;
}
int32_t zkhip_setup_gm17(zkhip_ctx* ctx, const zkhip_r1cs* r1cs, const uint8_t* toxic, const uint8_t* g1, const uint8_t* g2, uint8_t* pk_out,
                         uint64_t pk_cap) {
    if (!ctx) return ZKHIP_ERR_BAD_ARG;
    return guarded(ctx, [&] {
        require(r1cs && toxic && pk_out, ZKHIP_ERR_BAD_ARG, "null argument");
        require(r1cs->ctx == ctx, ZKHIP_ERR_BAD_ARG, "constraint system belongs to another context");
        ops_for(r1cs->curve)->gm17_setup(ctx, r1cs, toxic, g1, g2, pk_out, pk_cap);
    });
}

}  // extern "C"

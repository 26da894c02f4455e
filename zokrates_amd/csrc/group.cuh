// group.cuh — host-side launchers of the elliptic-curve kernels, instantiated once per (curve, group) in a
// translation unit of their own (bn254_g1.hip, bn254_g2.hip, bls381_g1.hip, bls381_g2.hip).
#pragma once
#include "core.cuh"
#include "bind.cuh"

namespace zk {

template <class FS>
void points_to_packed(zkhip_ctx* ctx, const Aff<FS>* d_in, void* d_out, u64 n) {
    typedef typename Unsat<FS>::type U;
    ZK_LAUNCH((k_points_to_packed<FS, U>), dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_in, (AffPacked<U>*)d_out, n);
}

// levels 1 .. W-1 of a base table whose level 0 (count points) is in place; level j starts at j * count.  Bases are
// processed in chunks so that the transient XYZZ / prefix-product workspace stays below ~1 GiB whatever the key size.
template <class FS>
void msm_table_levels(zkhip_ctx* ctx, void* d_table, u64 count, int c, int W) {
    typedef typename Unsat<FS>::type F;
    if (W <= 1 || count == 0) return;
    const u64 chunk = std::min<u64>(count, std::max<u64>(1024, ((u64)768 << 20) / ((u64)(W - 1) * (sizeof(Xyzz<F>) + sizeof(F)))));
    DBuf tmp, pre;
    tmp.ensure((u64)(W - 1) * chunk * sizeof(Xyzz<F>));
    pre.ensure((u64)(W - 1) * chunk * sizeof(F));
    for (u64 i0 = 0; i0 < count; i0 += chunk) {
        const u64 cnt = std::min(chunk, count - i0);
        ZK_LAUNCH((k_msm_table_levels<F, FS>), dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, (AffPacked<F>*)d_table, count, i0, cnt, c, W,
                  ptr<Xyzz<F>>(tmp), ptr<F>(pre));
    }
    stream_sync(ctx->stream);
}

// `nt` MSMs over ONE sorted list (tables[t] -> d_window_sums + t * sum_stride): one slicing, one accumulation launch, one
// fold — A, B1 and L of a proof share the sort of the assignment, and as one launch their residency is what the grid says
// (three separate accumulations pack five waves per SIMD and leave no room for a 128-register fold wave).
template <class FS>
void msm_run_tables(zkhip_ctx* ctx, MsmLane& lane, const MsmSort& so, const void* const* d_tables, int nt, const MsmShape& sh, Xyzz<FS>* d_window_sums,
                    u32 sum_stride, Event ev_begin, Event ev_end, Event accum_after, Xyzz<FS>* h_window_sums) {
    typedef typename Unsat<FS>::type F;   // the kernels run on the unsaturated field
    require(nt >= 1 && nt <= MSM_MAX_TABLES, ZKHIP_ERR_BAD_ARG, "internal: number of tables of one MSM launch");
    MsmTables tables{};
    for (int t = 0; t < nt; ++t) tables.p[t] = d_tables[t];
    Stream s = ctx->serial ? ctx->stream : lane_stream(lane);
    stream_wait_event(s, so.ready);
    // one slice of the sorted list per work-item the machine holds (never finer than MSM_MIN_SLICE entries)
    // (the slices may outnumber the work-items the kernel's registers let the machine hold: SLICE_WPE >= ACCUM_WPE)
    const int single = MsmTuning<F>::IS_EXT ? (ctx->msm_g2_waves ? ctx->msm_g2_waves : MsmTuning<F>::SLICE_WPE)
                                            : (ctx->msm_g1_waves ? ctx->msm_g1_waves : MsmTuning<F>::SLICE_WPE);
    const int wpe = ctx->msm_waves ? ctx->msm_waves : (nt > 1 ? std::max(1, (ctx->msm_fused_waves ? ctx->msm_fused_waves : MsmTuning<F>::FUSED_WPE)) : single);
    const u64 machine = ctx->msm_lanes ? (u64)ctx->msm_lanes : (u64)ctx->cus * 4 * 64 * wpe / (nt > 1 && !ctx->msm_waves ? nt : 1);
    const u32 nlanes = (u32)std::max<u64>(1, std::min<u64>(machine, (sh.n * (u64)sh.W + ctx->msm_min_slice - 1) / ctx->msm_min_slice));
    const bool share = lane.share_cu && !ctx->serial;
    lane.share_cu = false;
    const bool lone_launch = lane.lone_launch;
    lane.lone_launch = false;
    const MsmCut cut{nlanes, ctx->msm_min_slice, (u32)std::min<u64>(sh.n * (u64)sh.levels, 0x7fffffffu), share ? 1u : 0u};
    const u64 partial_stride = (u64)sh.nkeys + nlanes;
    lane.heavy.ensure(((size_t)sh.nkeys + 1) * 4);            // [0] = count, [1..] = keys
    lane.lane_key.ensure((size_t)nlanes * 4);
    lane.partial.ensure((size_t)nt * partial_stride * sizeof(Xyzz<F>));
    lane.bucket.ensure((size_t)nt * sh.nkeys * sizeof(Xyzz<F>));
    lane.rows.ensure((size_t)nt * sh.sets * sh.H * sizeof(Xyzz<F>));
    lane.cols.ensure((size_t)nt * sh.sets * sh.Lw * sizeof(Xyzz<F>));
    lds_opt_in(ctx, (const void*)k_msm_fold_rows<F>);
    lds_opt_in(ctx, (const void*)k_msm_fold_cols<F>);
    lds_opt_in(ctx, (const void*)k_msm_fold_final<F, FS>);
    lds_opt_in(ctx, (const void*)k_msm_fold_final_scan<F, FS>);
    const unsigned T = 256;
    ZK_LAUNCH(k_msm_lane_keys, dim3(blocks_for(nlanes, T)), dim3(T), 0, s, ptr<u32>(so.off), sh.nkeys, cut, ptr<u32>(lane.lane_key), ptr<u32>(lane.heavy));
    ZK_LAUNCH(k_msm_find_heavy, dim3(blocks_for(sh.nkeys, T)), dim3(T), 0, s, ptr<u32>(so.off), sh.nkeys, cut, ptr<u32>(lane.heavy) + 1,
              ptr<u32>(lane.heavy));
    if (accum_after && !ctx->serial) stream_wait_event(s, accum_after);   // (the slicing above only needs the sort)
    if (ev_begin) event_record(ev_begin, s);
    // (share: padded beyond half of the CU's 160 KiB, so that a second workgroup of this launch does not fit beside the first — the
    // other half of the LDS and of the registers stays free for whatever else arrives)
    const size_t acc_lds = share ? std::max<size_t>(msm_accum_lds_bytes<F>(), (size_t)84 * 1024) : msm_accum_lds_bytes<F>();
    if (ctx->skip_inf_mode == 1 || (ctx->skip_inf_mode == 0 && sh.skip_inf)) {
        if (acc_lds > 64 * 1024) lds_opt_in(ctx, (const void*)k_msm_accum<F, MsmTuning<F>::ACCUM_WPE, true>);
        ZK_LAUNCH((k_msm_accum<F, MsmTuning<F>::ACCUM_WPE, true>), dim3(blocks_for(nlanes, T), nt), dim3(T), acc_lds, s, tables, ptr<u32>(so.off), ptr<u32>(so.sorted),
                  ptr<u32>(lane.lane_key), ptr<Xyzz<F>>(lane.partial), partial_stride, sh.nkeys, cut);
    } else {
        if (acc_lds > 64 * 1024) lds_opt_in(ctx, (const void*)k_msm_accum<F, MsmTuning<F>::ACCUM_WPE, false>);
        ZK_LAUNCH((k_msm_accum<F, MsmTuning<F>::ACCUM_WPE, false>), dim3(blocks_for(nlanes, T), nt), dim3(T), acc_lds, s, tables, ptr<u32>(so.off), ptr<u32>(so.sorted),
                  ptr<u32>(lane.lane_key), ptr<Xyzz<F>>(lane.partial), partial_stride, sh.nkeys, cut);
    }
    if (ev_end) event_record(ev_end, s);
    // the fold chain on another stream (hardware queue) than the accumulation: a lone proof's — only where the stream plan made the lane
    // a lone fold stream (the hop is an event on the proof's critical path); a batch's — the lane's fold stream of the plan, or zkhip_ctx::fold_hop
    // (a lone proof's G2 lane — the first to finish, its fold chain the longest — also takes the lane's batch fold stream: 10.3-10.9 ms without, 9.7-9.9 with)
    const bool lone_hop = lone_launch && (lane.lone_fold_made || (MsmTuning<F>::IS_EXT && lane.fold_made));
    if (!ctx->serial && (lone_launch ? lone_hop : (lane.fold_made || ctx->fold_hop == 1 || (ctx->fold_hop == 2 && MsmTuning<F>::IS_EXT)))) {
        event_record(lane.acc_done, s);
        s = (lone_launch && lane.lone_fold_made) ? lane.lone_fold_stream : lane_fold_stream(lane);
        stream_wait_event(s, lane.acc_done);
    }
    if (ctx->heavy_runs) {
        lds_opt_in(ctx, (const void*)k_msm_heavy_reduce<F>);
        const unsigned TH = (unsigned)ctx->heavy_threads;
        ZK_LAUNCH((k_msm_heavy_reduce<F>), dim3(MSM_HEAVY_CHUNKS, nt), dim3(TH), (size_t)TH * sizeof(Xyzz<F>), s, ptr<Xyzz<F>>(lane.partial), partial_stride,
                  ptr<u32>(so.off), sh.nkeys, cut, ptr<u32>(lane.heavy) + 1, ptr<u32>(lane.heavy));
    }
    FoldDigits digs{};
    const u32 widest = std::max(sh.H, sh.Lw);
    if ((ctx->fold_lines == 1 || (ctx->fold_lines == 2 && nt == 1)) && widest <= 256) {
        // rows and columns of the bucket matrix in ONE launch (kernels_msm.cuh 5a')
        const unsigned TL = std::max<u32>(64, widest);
        lds_opt_in(ctx, (const void*)k_msm_fold_lines<F>);
        ZK_LAUNCH((k_msm_fold_lines<F>), dim3(widest, sh.sets * 2, nt), dim3(TL), (size_t)TL * sizeof(Xyzz<F>), s, ptr<Xyzz<F>>(lane.partial), partial_stride,
                  ptr<u32>(so.off), sh.nkeys, cut, sh.K, sh.Lw, sh.H, ptr<u32>(lane.heavy) + 1, ptr<u32>(lane.heavy), ctx->heavy_runs ? 1u : 0u,
                  ptr<Xyzz<F>>(lane.rows), ptr<Xyzz<F>>(lane.cols));
    } else {
        ZK_LAUNCH((k_msm_fold_rows<F>), dim3(sh.H, sh.sets, nt), dim3(sh.Lw), (size_t)sh.Lw * sizeof(Xyzz<F>), s, ptr<Xyzz<F>>(lane.partial), partial_stride,
                  ptr<u32>(so.off), sh.nkeys, cut, sh.K, sh.Lw, ptr<u32>(lane.heavy) + 1, ptr<u32>(lane.heavy), ctx->heavy_runs ? 1u : 0u, ptr<Xyzz<F>>(lane.bucket),
                  ptr<Xyzz<F>>(lane.rows));
        // column sums of the K = H x Lw buckets of a set; work-item (lo, hg) adds RG / HG rows serially
        const u32 RG = sh.H;
        const u32 HG = std::max<u32>(1, std::min<u32>((u32)ctx->fold_hg, RG)), CW = std::min<u32>(sh.Lw, 256 / HG), NG = sh.H / RG;
        ZK_LAUNCH((k_msm_fold_cols<F>), dim3(sh.Lw / CW, sh.sets * NG, nt), dim3(CW, HG), (size_t)CW * HG * sizeof(Xyzz<F>), s, ptr<Xyzz<F>>(lane.bucket), (u64)sh.K,
                  (u64)sh.nkeys, sh.Lw, sh.H, RG, ptr<Xyzz<F>>(lane.cols));
    }
    digs.d[0] = FoldDigit{lane.cols.p, sh.Lw, 1, 0};
    digs.d[1] = FoldDigit{lane.rows.p, sh.H, 0, (u32)ilog2_floor(sh.Lw)};
    // the scan form of the last fold step: one workgroup of <= 256 work-items per digit; the double-and-add form (two digits
    // in one workgroup of Lw work-items) is the fallback.  Either leaves one sum per digit and bucket set.
    u32 longest = 0;
    for (u32 d = 0; d < sh.ndig; ++d) longest = std::max(longest, digs.d[d].len);
    const unsigned TS = std::max<u32>(64, longest);
    if (ctx->fold_scan && TS <= 256) {
        ZK_LAUNCH((k_msm_fold_final_scan<F, FS>), dim3(sh.sets, nt, sh.ndig), dim3(TS), TS * sizeof(Xyzz<F>), s, digs, d_window_sums, sum_stride);
    } else {
        const unsigned TF = std::max<u32>(64, sh.Lw);
        ZK_LAUNCH((k_msm_fold_final<F, FS>), dim3(sh.sets, nt), dim3(TF), TF * sizeof(Xyzz<F>), s, ptr<Xyzz<F>>(lane.rows), ptr<Xyzz<F>>(lane.cols), sh.Lw,
                  sh.H, d_window_sums, sum_stride);
    }
    if (h_window_sums)          // the lane's own copy-out (pinned host memory of the proof slot: truly asynchronous)
        for (int t = 0; t < nt; ++t)
            dev_d2h_pinned(h_window_sums + (size_t)t * sum_stride, d_window_sums + (size_t)t * sum_stride, (size_t)sh.nsums() * sizeof(Xyzz<FS>), s);
    event_record(lane.done, s);
}
// number of points at infinity among the first `count` entries (level 0) of a packed table; synchronises ctx->stream
template <class FS>
u64 count_infinite(zkhip_ctx* ctx, const void* d_table, u64 count) {
    typedef typename Unsat<FS>::type F;
    if (!count) return 0;
    DBuf d;
    d.ensure(4);
    dev_memset(d.p, 0, 4, ctx->stream);
    ZK_LAUNCH((k_count_infinite<F>), dim3(blocks_for(count, 256)), dim3(256), 0, ctx->stream, (const AffPacked<F>*)d_table, count, ptr<u32>(d));
    u32 n = 0;
    dev_d2h(&n, d.p, 4, ctx->stream);
    stream_sync(ctx->stream);
    return n;
}
// sets, in a bitmap over the first `count` entries (level 0) of a packed table, the bit of every FINITE point; on ctx->stream
template <class FS>
void mark_finite(zkhip_ctx* ctx, const void* d_table, u64 count, u32* d_bitmap) {
    typedef typename Unsat<FS>::type F;
    ZK_LAUNCH((k_mark_finite<F>), dim3(blocks_for(count, 256)), dim3(256), 0, ctx->stream, (const AffPacked<F>*)d_table, count, d_bitmap);
}
template <class FS>
void msm_run(zkhip_ctx* ctx, MsmLane& lane, const MsmSort& so, const void* d_table, const MsmShape& sh, Xyzz<FS>* d_window_sums,
             Event ev_begin, Event ev_end, Event accum_after, Xyzz<FS>* h_window_sums) {
    msm_run_tables<FS>(ctx, lane, so, &d_table, 1, sh, d_window_sums, 0, ev_begin, ev_end, accum_after, h_window_sums);
}

template <class F>
void fixed_base_table(zkhip_ctx* ctx, const Aff<F>* h_pj, int nwin, DBuf& tbl) {
    DBuf d_pj;
    d_pj.ensure((size_t)nwin * sizeof(Aff<F>));
    dev_h2d(d_pj.p, h_pj, (size_t)nwin * sizeof(Aff<F>), ctx->stream);
    tbl.ensure((size_t)nwin * 256 * sizeof(Aff<F>));
    ZK_LAUNCH((k_fixed_base_table<F>), dim3(blocks_for(nwin * 256, 64)), dim3(64), 0, ctx->stream, ptr<Aff<F>>(d_pj), ptr<Aff<F>>(tbl), nwin);
    stream_sync(ctx->stream);
}
template <class F>
void fixed_base_mul(zkhip_ctx* ctx, const DBuf& tbl, int nwin, const u32* d_scalars, u64 count, Aff<F>* d_out) {
    ZK_LAUNCH((k_fixed_base_mul<F>), dim3(blocks_for(count, 64)), dim3(64), 0, ctx->stream, d_scalars, count, ptr<Aff<F>>(tbl), nwin, d_out);
}

// ---- binding a key to a constraint system (bind.cuh): launchers, G1 only ----
template <class FS>
size_t bind_xyzz_bytes() { return sizeof(Xyzz<typename Unsat<FS>::type>); }
template <class FS>
void bind_scale(zkhip_ctx* ctx, const void* d_h_table, u64 N, u64 n_src, u32 n1, u32 n2, u32 n3, const u32* d_scal, const u32* d_konst, int nw, void* d_x) {
    typedef typename Unsat<FS>::type F;
    ZK_LAUNCH((k_bind_scale<F>), dim3(blocks_for(N, 256), 2), dim3(256), 0, ctx->stream, (const AffPacked<F>*)d_h_table, N, n_src, n1, n2, n3, d_scal, d_konst, nw,
              (Xyzz<F>*)d_x);
}
template <class FS>
void bind_fft(zkhip_ctx* ctx, void* d_x, u64 N, const u32* d_tw, int nw) {
    typedef typename Unsat<FS>::type F;
    for (u64 q = N / 2; q >= 1; q >>= 1)
        ZK_LAUNCH((k_bind_fft_stage<F>), dim3(blocks_for(N / 2, 256), 2), dim3(256), 0, ctx->stream, (Xyzz<F>*)d_x, N, q, d_tw, nw);
}
template <class FS>
void bind_h_finish(zkhip_ctx* ctx, const void* d_x, u64 N, int logN, void* d_out) {
    typedef typename Unsat<FS>::type F;
    ZK_LAUNCH((k_bind_h_finish<F>), dim3(blocks_for(N, 256)), dim3(256), 0, ctx->stream, (const Xyzz<F>*)d_x, N, logN, (AffPacked<F>*)d_out);
}
template <class FS>
void bind_cmul(zkhip_ctx* ctx, const void* d_x, int logN, const u32* d_row, const u32* d_val, int nw, const u32* d_minus_one, u64 nnz, void* d_prod) {
    typedef typename Unsat<FS>::type F;
    ZK_LAUNCH((k_bind_cmul<F>), dim3(blocks_for(nnz, 256)), dim3(256), 0, ctx->stream, (const Xyzz<F>*)d_x, logN, d_row, d_val, nw, d_minus_one, nnz, (Xyzz<F>*)d_prod);
}
template <class FS>
void bind_l_finish(zkhip_ctx* ctx, const void* d_prod, const u64* d_cptr, const void* d_l_table, u64 m, const u32* d_long_cols, u64 n_long, void* d_sum, void* d_out) {
    typedef typename Unsat<FS>::type F;
    ZK_LAUNCH((k_bind_l_sum_short<F>), dim3(blocks_for(m, 256)), dim3(256), 0, ctx->stream, (const Xyzz<F>*)d_prod, d_cptr, (const AffPacked<F>*)d_l_table, m, (Xyzz<F>*)d_sum);
    if (n_long)
        ZK_LAUNCH((k_bind_l_sum_long<F>), dim3((unsigned)n_long), dim3(64), 0, ctx->stream, (const Xyzz<F>*)d_prod, d_cptr, (const AffPacked<F>*)d_l_table, d_long_cols, (Xyzz<F>*)d_sum);
    ZK_LAUNCH((k_bind_l_affine<F>), dim3(blocks_for(m, 256)), dim3(256), 0, ctx->stream, (const Xyzz<F>*)d_sum, m, (AffPacked<F>*)d_out);
}
#define ZK_INSTANTIATE_BIND(F)                                                                                          \
    template size_t bind_xyzz_bytes<F>();                                                                               \
    template void bind_scale<F>(zkhip_ctx*, const void*, u64, u64, u32, u32, u32, const u32*, const u32*, int, void*);  \
    template void bind_fft<F>(zkhip_ctx*, void*, u64, const u32*, int);                                                 \
    template void bind_h_finish<F>(zkhip_ctx*, const void*, u64, int, void*);                                           \
    template void bind_cmul<F>(zkhip_ctx*, const void*, int, const u32*, const u32*, int, const u32*, u64, void*);      \
    template void bind_l_finish<F>(zkhip_ctx*, const void*, const u64*, const void*, u64, const u32*, u64, void*, void*);

#define ZK_INSTANTIATE_GROUP(F)                                                                                         \
    template void msm_run<F>(zkhip_ctx*, MsmLane&, const MsmSort&, const void*, const MsmShape&, Xyzz<F>*, Event, Event, Event, Xyzz<F>*);   \
    template void msm_run_tables<F>(zkhip_ctx*, MsmLane&, const MsmSort&, const void* const*, int, const MsmShape&, Xyzz<F>*, u32, Event, Event, Event, Xyzz<F>*); \
    template void points_to_packed<F>(zkhip_ctx*, const Aff<F>*, void*, u64);                    \
    template void msm_table_levels<F>(zkhip_ctx*, void*, u64, int, int);                         \
    template u64 count_infinite<F>(zkhip_ctx*, const void*, u64);                                \
    template void mark_finite<F>(zkhip_ctx*, const void*, u64, u32*);                            \
    template void fixed_base_table<F>(zkhip_ctx*, const Aff<F>*, int, DBuf&);                                           \
    template void fixed_base_mul<F>(zkhip_ctx*, const DBuf&, int, const u32*, u64, Aff<F>*);

}  // namespace zk

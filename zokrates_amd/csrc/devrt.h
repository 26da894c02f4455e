// devrt.h — the thin device-runtime layer under libzkhip.
//
// Product build (hipcc, gfx950): straight HIP — hipMalloc / hipMemcpyAsync / streams / events and
// `hipLaunchKernelGGL`.  There is NO CPU fallback in the product: if no GPU is present every entry
// point of libzkhip.so fails with ZKHIP_ERR_DEVICE.
//
// Test-only build (-DZK_EMU, g++): the same kernel sources run on a single-threaded fibre
// emulator (emu.h) so that indexing / LDS / barrier logic can be exercised by `pytest -m "not gpu"`
// in a container without a GPU.  That build produces tests/_emu/libzkhip_emu.so, never
// libzkhip.so, and is loaded only by tests.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <mutex>
#include <unordered_map>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#ifdef ZK_EMU
#include "emu.h"
#else
#include <hip/hip_runtime.h>
#endif

namespace zk {

struct DevError {
    std::string msg;
};

// A few host threads that are always joined: if starting one fails (EAGAIN) the work runs on the calling thread instead,
// and leaving the scope — normally or through an exception — joins whatever was started.
struct HostThreads {
    std::vector<std::thread> th;
    HostThreads() { th.reserve(64); }
    HostThreads(const HostThreads&) = delete;
    HostThreads& operator=(const HostThreads&) = delete;
    template <class Fn>
    void run(Fn&& fn) {
        try {
            th.emplace_back(fn);
        } catch (const std::system_error&) {
            fn();
        }
    }
    void join() {
        for (auto& t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
    ~HostThreads() { join(); }
};


// [0, n) cut into a few contiguous ranges, one host thread each (memory-bound host loops: key decoding)
template <class Fn>
inline void host_parallel_for(uint64_t n, Fn&& fn, unsigned max_threads = 8) {
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, max_threads), n / 65536));
    if (T <= 1) { fn((uint64_t)0, n); return; }
    HostThreads th;
    for (unsigned t = 1; t < T; ++t) th.run([&fn, n, t, T] { fn(n * t / T, n * (t + 1) / T); });
    fn((uint64_t)0, n / T);
    th.join();
}

#ifndef ZK_EMU
#define ZK_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw zk::DevError{std::string(#expr) + ": " + hipGetErrorString(e_)};                \
    } while (0)

typedef hipStream_t Stream;
typedef hipEvent_t Event;

inline int dev_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
inline void dev_set(int d) { ZK_HIP_CHECK(hipSetDevice(d)); }
inline void dev_pci_bus_id(int d, char* out, size_t cap) { ZK_HIP_CHECK(hipDeviceGetPCIBusId(out, (int)cap, d)); }
// Device allocations.  Two measured effects on MI355X, both on the latency-bound fold kernels (same device code, run times
// differing by 2x between allocation patterns):
//  * buffers whose sizes are not multiples of 2 MiB can end up on small pages -> every allocation is a whole number of
//    2 MiB pages;
//  * buffers that all START on a 2 MiB boundary alias each other in the caches / memory channels (the fold kernels walk
//    five or six of them in lock step) -> each allocation is handed out at a different offset (a multiple of 4352 B =
//    17 x 256 B) inside its first page.
struct DevAllocTable {
    std::mutex mu;
    std::unordered_map<void*, void*> base_of;   // pointer handed out -> pointer hipMalloc returned
    unsigned counter = 0;
};
inline DevAllocTable& dev_alloc_table() {
    static DevAllocTable t;
    return t;
}
inline void* dev_alloc(size_t bytes) {
    const size_t page = (size_t)2 << 20, slots = 61;
    static const size_t step = getenv("ZKHIP_ALLOC_STEP") ? (size_t)atol(getenv("ZKHIP_ALLOC_STEP")) / 256 * 256 : 4352;
    DevAllocTable& t = dev_alloc_table();
    std::lock_guard<std::mutex> lock(t.mu);
    const size_t offset = (size_t)(t.counter++ % slots) * step;
    void* base = nullptr;
    ZK_HIP_CHECK(hipMalloc(&base, (std::max<size_t>(bytes, 1) + offset + page - 1) / page * page));
    void* p = (char*)base + offset;
    t.base_of[p] = base;
    return p;
}
inline void dev_free(void* p) {
    if (!p) return;
    DevAllocTable& t = dev_alloc_table();
    std::lock_guard<std::mutex> lock(t.mu);
    auto it = t.base_of.find(p);
    if (it == t.base_of.end()) return;
    (void)hipFree(it->second);
    t.base_of.erase(it);
}
inline void* host_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    ZK_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault));
    return p;
}
inline void host_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}
// Race-hunting mode (ZKHIP_TUNE_STREAM_JITTER / ZKHIP_STREAM_JITTER=<max microseconds> in the environment when the library
// is loaded).  The library drives ~20 streams per context and orders them with events; an ordering that is missing is
// invisible as long as the producer happens to win, and the emulator (one synchronous fibre scheduler) cannot see it at
// all.  With jitter on, every enqueue is preceded — with probability 1/2 — by a one-wave spin kernel of random length on
// the same stream: producers are delayed relative to consumers that do not wait for them, consumers that do wait are
// unaffected, and a result that depended on luck changes.
struct JitterState {
    std::atomic<int> max_us{0};
    std::atomic<uint64_t> rng{0x9E3779B97F4A7C15ull};
};
inline JitterState& jitter_state() {
    static JitterState st;
    static const bool init = [] {
        if (const char* e = getenv("ZKHIP_STREAM_JITTER")) st.max_us.store(std::max(0, std::min(5000, atoi(e))));
        return true;
    }();
    (void)init;
    return st;
}
static __global__ void k_jitter_spin(unsigned long long ticks) {   // wall_clock64: the constant 100 MHz counter
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
inline void jitter_before(Stream s) {
    JitterState& st = jitter_state();
    const int max_us = st.max_us.load(std::memory_order_relaxed);
    if (max_us <= 0) return;
    uint64_t x = st.rng.fetch_add(0x9E3779B97F4A7C15ull, std::memory_order_relaxed);   // SplitMix64 step
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    if (x & 1) return;
    const unsigned long long us = (x >> 8) % (uint64_t)(max_us + 1);
    hipLaunchKernelGGL(k_jitter_spin, dim3(1), dim3(1), 0, s, us * 100ull);
}
// Copies between the device and host memory THE CALLER OWNS (pageable: numpy arrays, std::vector, stack).  The HIP runtime
// accepts such pointers in hipMemcpyAsync by pinning the user's pages behind the caller's back for the duration of the
// transfer; on this stack that path faults now and then when several transfers and kernels are queued without a host
// synchronisation between them (round 2's driver run died that way, and removing an unrelated 40 ms host loop between the
// three uploads of zkhip_r1cs_load made it die every time: profiles/r3c_pageable_copy_fault.md).  So the library never
// hands a pointer it does not own to an asynchronous copy: pageable data goes through a pinned staging ring that belongs to
// the library, in chunks — host memcpy into the ring, asynchronous copy out of it, an event per chunk before its reuse.
// dev_h2d / dev_d2h are therefore safe for ANY host pointer; dev_h2d_pinned / dev_d2h_pinned are the raw asynchronous copies
// for memory the library allocated with host_alloc_pinned.
inline int& copy_mode() {
    static int mode = getenv("ZKHIP_COPY_MODE") ? atoi(getenv("ZKHIP_COPY_MODE")) : 2;   // 0: raw async (the runtime pins), 1: hipMemcpy, 2: staged
    return mode;
}
// host threads that fill the ring for one transfer of two chunks and more (ZKHIP_COPY_THREADS, 1 .. 8; read once)
inline unsigned staging_copy_threads() {
    static const unsigned t = [] {
        const char* e = getenv("ZKHIP_COPY_THREADS");
        const int v = e ? atoi(e) : 4;
        const unsigned hw = std::thread::hardware_concurrency();
        return (unsigned)std::max(1, std::min(std::min(8, v), hw ? (int)hw : 1));
    }();
    return t;
}
struct StagingRing {
    static constexpr size_t CHUNK = (size_t)8 << 20;
    static constexpr int SLOTS = 4;
    void* buf[SLOTS] = {};
    hipEvent_t ev[SLOTS] = {};
    bool used[SLOTS] = {};
    int next = 0;
    int device = -1;
    std::mutex mu;
    void init() {
        if (buf[0]) return;
        for (int i = 0; i < SLOTS; ++i) {
            ZK_HIP_CHECK(hipHostMalloc(&buf[i], CHUNK, hipHostMallocDefault));
            ZK_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
    // a slot whose previous transfer has completed
    int acquire() {
        const int i = next;
        next = (next + 1) % SLOTS;
        if (used[i]) ZK_HIP_CHECK(hipEventSynchronize(ev[i]));
        used[i] = false;
        return i;
    }
};
// one ring per device (contexts on one device share it under its mutex; the copies of one call are issued under the lock)
inline StagingRing& staging_ring() {
    static StagingRing rings[64];
    int d = 0;
    (void)hipGetDevice(&d);
    StagingRing& r = rings[d & 63];
    return r;
}
// Before a context destroys its streams: wait for every transfer the ring still tracks and forget it.  An event keeps a
// reference to the stream it was last recorded on; synchronising on it after that stream is gone is undefined (HIP answered
// "operation not permitted on an event last recorded in a capturing stream" once: freed stream memory read as a capture flag).
inline void staging_drain() {
    StagingRing& r = staging_ring();
    std::lock_guard<std::mutex> lock(r.mu);
    if (!r.buf[0]) return;
    for (int i = 0; i < StagingRing::SLOTS; ++i)
        if (r.used[i]) {
            (void)hipEventSynchronize(r.ev[i]);
            r.used[i] = false;
        }
}
inline void dev_h2d_pinned(void* d, const void* h, size_t n, Stream s) { jitter_before(s); ZK_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); }
inline void dev_d2h_pinned(void* h, const void* d, size_t n, Stream s) { jitter_before(s); ZK_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); }
inline void dev_h2d(void* d, const void* h, size_t n, Stream s) {
    if (!n) return;
    const int mode = copy_mode();
    if (mode == 0) { dev_h2d_pinned(d, h, n, s); return; }
    if (mode == 1) {   // the synchronous API: ordered after what the stream already holds, complete when it returns
        ZK_HIP_CHECK(hipStreamSynchronize(s));
        ZK_HIP_CHECK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        return;
    }
    StagingRing& r = staging_ring();
    std::lock_guard<std::mutex> lock(r.mu);
    r.init();
    const size_t nchunks = (n + StagingRing::CHUNK - 1) / StagingRing::CHUNK;
    const unsigned T = nchunks >= 2 ? staging_copy_threads() : 1;
    if (T > 1) {
        // One thread copies ~15 GB/s into the ring: the 32 MiB assignment of a 2^20 proof that comes from host memory spent 2.2 ms
        // there with the DMA engine mostly idle.  A few helpers, started ONCE per transfer (starting them per chunk cost more than
        // it gained: profiles/r7g_*), each copy their share of every chunk; the calling thread copies share 0 and issues the DMA.
        struct Shared {
            std::atomic<long> go{-1}, done{0};
            std::atomic<bool> abort{false};
            unsigned T = 1;               // threads that copy (the caller included): set before the first `go`
            std::vector<char*> slot;
        } sh;
        sh.slot.assign(nchunks, nullptr);
        auto share = [n](size_t c, unsigned t, unsigned T_, size_t& lo, size_t& hi) {
            const size_t len = std::min(StagingRing::CHUNK, n - c * StagingRing::CHUNK);
            lo = len * t / T_ / 64 * 64;
            hi = t + 1 == T_ ? len : len * (t + 1) / T_ / 64 * 64;
        };
        struct Release {      // (an exception of the issuing thread must not leave the helpers waiting)
            Shared& sh;
            ~Release() { sh.abort.store(true, std::memory_order_release); }
        };
        HostThreads th;       // (joined after `release` has run: declared first, destroyed last)
        Release release{sh};
        for (unsigned t = 1; t < T; ++t) {
            try {
                const unsigned me = (unsigned)th.th.size() + 1;
                th.th.emplace_back([&sh, &share, h, nchunks, me] {
                    for (size_t c = 0; c < nchunks; ++c) {
                        while (sh.go.load(std::memory_order_acquire) < (long)c) {
                            if (sh.abort.load(std::memory_order_acquire)) return;
                            std::this_thread::yield();
                        }
                        size_t lo, hi;
                        share(c, me, sh.T, lo, hi);
                        memcpy(sh.slot[c] + lo, (const char*)h + c * StagingRing::CHUNK + lo, hi - lo);
                        sh.done.fetch_add(1, std::memory_order_release);
                    }
                });
            } catch (const std::system_error&) {      // (no thread to be had: the others share the work)
                break;
            }
        }
        const long helpers = (long)th.th.size();
        sh.T = (unsigned)helpers + 1;
        for (size_t c = 0; c < nchunks; ++c) {
            const size_t off = c * StagingRing::CHUNK, len = std::min(StagingRing::CHUNK, n - off);
            const int i = r.acquire();
            sh.slot[c] = (char*)r.buf[i];
            sh.go.store((long)c, std::memory_order_release);
            size_t lo, hi;
            share(c, 0, sh.T, lo, hi);
            memcpy(sh.slot[c] + lo, (const char*)h + off + lo, hi - lo);
            while (sh.done.load(std::memory_order_acquire) < (long)(c + 1) * helpers) std::this_thread::yield();
            jitter_before(s);
            ZK_HIP_CHECK(hipMemcpyAsync((char*)d + off, r.buf[i], len, hipMemcpyHostToDevice, s));
            ZK_HIP_CHECK(hipEventRecord(r.ev[i], s));
            r.used[i] = true;
        }
        return;
    }
    for (size_t off = 0; off < n; off += StagingRing::CHUNK) {
        const size_t len = std::min(StagingRing::CHUNK, n - off);
        const int i = r.acquire();
        memcpy(r.buf[i], (const char*)h + off, len);
        jitter_before(s);
        ZK_HIP_CHECK(hipMemcpyAsync((char*)d + off, r.buf[i], len, hipMemcpyHostToDevice, s));
        ZK_HIP_CHECK(hipEventRecord(r.ev[i], s));
        r.used[i] = true;
    }
}
// The same transfer for bytes that are PRODUCED rather than copied (a proving key decoded from ark's encoding, a key image read
// from a mapped file): fill(dst, off, len) writes bytes [off, off + len) of the transfer to dst — straight into the ring's
// pinned slot, on up to `threads` host threads per chunk — so the data makes one pass through host memory instead of three
// (decode into a vector, copy into the ring, DMA).  `unit`: fill is only called with off and len that are multiples of it.
template <class Fill>
inline void dev_h2d_fill(void* d, size_t n, size_t unit, Stream s, Fill&& fill, unsigned threads = 8) {
    if (!n) return;
    StagingRing& r = staging_ring();
    std::lock_guard<std::mutex> lock(r.mu);
    r.init();
    const size_t chunk = std::max<size_t>(unit, StagingRing::CHUNK / unit * unit);
    if (chunk > StagingRing::CHUNK) throw DevError{"dev_h2d_fill: unit larger than a staging slot"};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t len = std::min(chunk, n - off);
        const int i = r.acquire();
        char* dst = (char*)r.buf[i];
        const size_t units = len / unit;
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, units / 4096));
        if (T <= 1) {
            fill(dst, off, len);
        } else {
            HostThreads th;
            for (unsigned t = 1; t < T; ++t) {
                const size_t lo = units * t / T * unit, hi = (t + 1 == T ? len : units * (t + 1) / T * unit);
                th.run([&fill, dst, off, lo, hi] { fill(dst + lo, off + lo, hi - lo); });
            }
            fill(dst, off, units / T * unit);
            th.join();
        }
        jitter_before(s);
        ZK_HIP_CHECK(hipMemcpyAsync((char*)d + off, r.buf[i], len, hipMemcpyHostToDevice, s));
        ZK_HIP_CHECK(hipEventRecord(r.ev[i], s));
        r.used[i] = true;
    }
}
// (the host needs the bytes, so this one returns when they are there: every caller synchronised right after it anyway)
inline void dev_d2h(void* h, const void* d, size_t n, Stream s) {
    if (!n) return;
    const int mode = copy_mode();
    if (mode == 0) { dev_d2h_pinned(h, d, n, s); return; }
    if (mode == 1) {
        ZK_HIP_CHECK(hipStreamSynchronize(s));
        ZK_HIP_CHECK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
        return;
    }
    StagingRing& r = staging_ring();
    std::lock_guard<std::mutex> lock(r.mu);
    r.init();
    // two chunks in flight: while one is copied out of the ring on the host, the next one arrives
    int pending[2] = {-1, -1};
    size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
    int q = 0;
    auto drain = [&](int k) {
        if (pending[k] < 0) return;
        ZK_HIP_CHECK(hipEventSynchronize(r.ev[pending[k]]));
        memcpy((char*)h + pend_off[k], r.buf[pending[k]], pend_len[k]);
        r.used[pending[k]] = false;
        pending[k] = -1;
    };
    for (size_t off = 0; off < n; off += StagingRing::CHUNK) {
        const size_t len = std::min(StagingRing::CHUNK, n - off);
        drain(q);
        const int i = r.acquire();
        jitter_before(s);
        ZK_HIP_CHECK(hipMemcpyAsync(r.buf[i], (const char*)d + off, len, hipMemcpyDeviceToHost, s));
        ZK_HIP_CHECK(hipEventRecord(r.ev[i], s));
        r.used[i] = true;
        pending[q] = i; pend_off[q] = off; pend_len[q] = len;
        q ^= 1;
    }
    drain(q);
    drain(q ^ 1);
}
inline void dev_d2d(void* d, const void* s_, size_t n, Stream s) { jitter_before(s); ZK_HIP_CHECK(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s)); }
// device memory of `src_dev` -> device memory of `dst_dev` (the current device), on a stream of the latter: the members of a
// multi-GPU proof exchange their halves of the witness map this way (xGMI peer copy; the same device: a plain copy)
inline void dev_copy_between(void* d, int dst_dev, const void* s_, int src_dev, size_t n, Stream s) {
    jitter_before(s);
    if (dst_dev == src_dev) ZK_HIP_CHECK(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s));
    else ZK_HIP_CHECK(hipMemcpyPeerAsync(d, dst_dev, s_, src_dev, n, s));
}
inline void dev_memset(void* d, int v, size_t n, Stream s) { jitter_before(s); ZK_HIP_CHECK(hipMemsetAsync(d, v, n, s)); }
inline Stream stream_create() {
    Stream s;
    ZK_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
// highest scheduling priority the device offers (used for the short NTT pipeline, which otherwise queues behind the
// MSM streams' long kernels and delays the H MSM that depends on it)
inline Stream stream_create_high_priority() {
    int lo = 0, hi = 0;
    ZK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    Stream s;
    ZK_HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    return s;
}
inline Stream stream_create_low_priority() {
    int lo = 0, hi = 0;
    ZK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    Stream s;
    ZK_HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
    return s;
}
inline void stream_destroy(Stream s) { (void)hipStreamDestroy(s); }
inline void stream_sync(Stream s) { ZK_HIP_CHECK(hipStreamSynchronize(s)); }
inline void dev_sync_all() { (void)hipDeviceSynchronize(); }
// free / total bytes of the current device
inline void dev_mem_info(size_t* free_b, size_t* total_b) { ZK_HIP_CHECK(hipMemGetInfo(free_b, total_b)); }
inline Event event_create() {
    Event e;
    ZK_HIP_CHECK(hipEventCreate(&e));
    return e;
}
inline void event_destroy(Event e) { (void)hipEventDestroy(e); }
inline void event_record(Event e, Stream s) { ZK_HIP_CHECK(hipEventRecord(e, s)); }
inline void event_sync(Event e) { ZK_HIP_CHECK(hipEventSynchronize(e)); }
inline void stream_wait_event(Stream s, Event e) { ZK_HIP_CHECK(hipStreamWaitEvent(s, e, 0)); }
// Both events have been recorded; the caller has waited for the work it depends on, but an event on ANOTHER stream (the
// main stream's "MSMs issued" marker, say) may still sit in a hardware queue behind a different context's kernels —
// streams outnumber hardware queues — so "not ready" is answered by waiting, not by failing the proof.
inline float event_elapsed_ms(Event a, Event b) {
    float ms = 0;
    hipError_t e = hipEventElapsedTime(&ms, a, b);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        ZK_HIP_CHECK(hipEventSynchronize(a));
        ZK_HIP_CHECK(hipEventSynchronize(b));
        e = hipEventElapsedTime(&ms, a, b);
    }
    ZK_HIP_CHECK(e);
    return ms;
}
inline void dev_check_last() { ZK_HIP_CHECK(hipGetLastError()); }

// ZKHIP_TRACE=1 in the environment: synchronise after every launch and name it on stderr (debugging aid)
inline bool trace_enabled() {
    static const bool on = getenv("ZKHIP_TRACE") != nullptr;
    return on;
}
// (an empty grid — a key or program with a zero-length vector — is a no-op, as it is on the test emulator; HIP rejects it)
inline bool grid_nonempty(const dim3& g) { return g.x != 0 && g.y != 0 && g.z != 0; }
#define ZK_LAUNCH(kernel, grid, block, smem, stream, ...)                                 \
    do {                                                                                  \
        if (!zk::grid_nonempty(dim3(grid))) break;                                        \
        if (zk::trace_enabled()) fprintf(stderr, "[zkhip] launch %s ...", #kernel);       \
        zk::jitter_before(stream);                                                        \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);               \
        zk::dev_check_last();                                                             \
        if (zk::trace_enabled()) {                                                        \
            ZK_HIP_CHECK(hipStreamSynchronize(stream));                                   \
            fprintf(stderr, " done\n");                                                   \
        }                                                                                 \
    } while (0)
#define ZK_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// Issue priority of a wave (s_setprio, 0..3).  An accumulation kernel saturates the integer multiplier of every SIMD it
// sits on and the arbiter serves the OLDEST wave first, so a short kernel that arrives beside it only gets the cycles it
// leaves (kernel traces: a 0.14 ms mat-vec took 4 ms, a 0.05 ms bucket reduction 3.5 ms).  The short, latency-bound
// kernels of a proof — sort, scans, folds, mat-vec, transforms — raise their priority first thing; the accumulation
// kernels stay at 0.
#ifndef ZK_WAVE_PRIO
#define ZK_WAVE_PRIO 3
#endif
#define ZK_PRIO_HIGH() __builtin_amdgcn_s_setprio(ZK_WAVE_PRIO)
// true if the predicate holds in ANY active lane of the wavefront: a wave-uniform condition (a scalar branch, no exec masking —
// both sides of an `if` on it are ordinary control flow whose results meet without per-lane copies)
#define ZK_WAVE_ANY(pred) (__builtin_amdgcn_ballot_w64(pred) != 0)
// -DZK_CHECKED: a debugging build whose kernels check every index they compute from data another kernel produced (sorted
// entries, bucket offsets, partial slots, matrix columns) and trap instead of touching memory outside their buffers; with
// ZKHIP_TRACE=1 (synchronise after every launch) the launch that trapped is the last one named on stderr.
// tools/build_variant.sh checked -DZK_CHECKED; the parity suite must run clean on it (profiles/r3k_checked_build.log).
#ifdef ZK_CHECKED
#define ZK_ASSERT_IDX(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define ZK_ASSERT_IDX(cond) ((void)0)
#endif

#else  // ---------------------------- emulator ----------------------------

typedef int Stream;
typedef double* Event;
inline int dev_count() { return 1; }
inline void dev_set(int) {}
inline void dev_pci_bus_id(int, char* out, size_t cap) { snprintf(out, cap, "0000:00:00.0"); }   // (the emulator's one "device")
inline void* dev_alloc(size_t bytes) { return aligned_alloc(256, ((bytes ? bytes : 16) + 255) / 256 * 256); }
inline void dev_free(void* p) { free(p); }
inline void* host_alloc_pinned(size_t bytes) { return dev_alloc(bytes); }
inline void host_free_pinned(void* p) { free(p); }
inline void dev_h2d(void* d, const void* h, size_t n, Stream) { memcpy(d, h, n); }
template <class Fill>
inline void dev_h2d_fill(void* d, size_t n, size_t unit, Stream, Fill&& fill, unsigned = 8) {
    // (the emulator has no ring: the same fill, cut at a few arbitrary unit boundaries so that the offsets are exercised)
    const size_t units = n / unit, cut = units / 3 * unit;
    if (cut) fill((char*)d, (size_t)0, cut);
    if (n > cut) fill((char*)d + cut, cut, n - cut);
}
inline void dev_d2h(void* h, const void* d, size_t n, Stream) { memcpy(h, d, n); }
inline void staging_drain() {}
inline void dev_h2d_pinned(void* d, const void* h, size_t n, Stream) { memcpy(d, h, n); }
inline void dev_d2h_pinned(void* h, const void* d, size_t n, Stream) { memcpy(h, d, n); }
inline void dev_d2d(void* d, const void* s_, size_t n, Stream) { memcpy(d, s_, n); }
inline void dev_copy_between(void* d, int, const void* s_, int, size_t n, Stream) { memcpy(d, s_, n); }
inline void dev_memset(void* d, int v, size_t n, Stream) { memset(d, v, n); }
inline Stream stream_create() { return 0; }
inline Stream stream_create_high_priority() { return 0; }
inline Stream stream_create_low_priority() { return 0; }
inline void stream_destroy(Stream) {}
inline void stream_sync(Stream) {}
inline void dev_sync_all() {}
inline void dev_mem_info(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)1 << 40; }
inline Event event_create() { return new double(0); }
inline void event_destroy(Event e) { delete e; }
inline void event_record(Event e, Stream) { *e = emu::now_ms(); }
inline void event_sync(Event) {}
inline void stream_wait_event(Stream, Event) {}
inline float event_elapsed_ms(Event a, Event b) { return (float)(*b - *a); }
inline void dev_check_last() {}
struct JitterState { std::atomic<int> max_us{0}; };   // accepted and ignored: the emulator has one synchronous "stream"
inline JitterState& jitter_state() { static JitterState st; return st; }

#define ZK_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (size_t)(smem), [&]() { kernel(__VA_ARGS__); })
#define ZK_DYN_SMEM(name) unsigned char* name = emu::dyn_smem()
#define ZK_PRIO_HIGH() ((void)0)
#define ZK_WAVE_ANY(pred) (pred)      /* (the fibres of the emulator decide one by one: both sides compute the same result) */
#ifdef ZK_CHECKED
#define ZK_ASSERT_IDX(cond) do { if (!(cond)) { fprintf(stderr, "ZK_ASSERT_IDX failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); abort(); } } while (0)
#else
#define ZK_ASSERT_IDX(cond) ((void)0)
#endif

#endif

}  // namespace zk

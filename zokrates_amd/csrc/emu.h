// emu.h — TEST-ONLY single-threaded fibre emulator for the HIP kernels in this directory.
//
// Compiled only with -DZK_EMU (g++), only into tests/_emu/libzkhip_emu.so, only loaded by
// `pytest -m "not gpu"` tests.  It exists because the build container has no GPU and GPU minutes
// are scarce: it lets the *same kernel source* (indexing, LDS staging, barriers, atomics) run
// on tiny inputs on the CPU.  It is NOT a product fallback: libzkhip.so is never built with it.
//
// Model: one workgroup at a time; each work-item is a ucontext fibre; `__syncthreads()` yields to
// the scheduler until every live fibre of the block has arrived; wave-level exchanges
// (`__shfl*`, `__ballot`) rendezvous the 64 fibres of a wave the same way.
#pragma once
#include <setjmp.h>
#include <ucontext.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }

namespace emu {

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    ucontext_t ctx;     // first entry only (gives the fibre its stack)
    jmp_buf jb;         // every later switch: _setjmp/_longjmp save no signal mask, i.e. make no system call
    int state;
    bool started;
    dim3 tid;
    unsigned flat;
};
struct Global {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    jmp_buf sched;
    Fiber* cur = nullptr;
    const std::function<void()>* fn = nullptr;
    unsigned char* smem = nullptr;
    size_t smem_cap = 0;
    char* stacks = nullptr;
    size_t stack_bytes = 0;
    unsigned max_fibers = 0;
    Fiber* fibers = nullptr;   // the current block's work-items
    unsigned nt = 0, done = 0;
    uint64_t xchg[64 * 32];   // per-wave exchange slots (up to 32 waves of 64 lanes)
};
inline Global& G() {
    static Global g;
    return g;
}
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline unsigned char* dyn_smem() { return G().smem; }

inline void yield_with(State s) {
    Global& g = G();
    Fiber* f = g.cur;
    f->state = s;
    if (_setjmp(f->jb) == 0) _longjmp(g.sched, 1);
}
inline void sync_block() { yield_with(WAIT_BLOCK); }
inline void sync_wave() { yield_with(WAIT_WAVE); }

// A fibre runs work-items back to back for as long as they finish without waiting: a work-item that returns hands its
// stack to the next one that has not started yet, so kernels without barriers cost one context switch per workgroup
// (a fresh context costs a system call; later switches are _setjmp/_longjmp), and a work-item that does wait simply leaves the chain suspended in its own slot.
inline void trampoline() {
    Global& g = G();
    for (;;) {
        (*g.fn)();
        Fiber* f = g.cur;
        f->state = DONE;
        ++g.done;
        const unsigned nx = f->flat + 1;
        if (nx < g.nt && !g.fibers[nx].started) {
            Fiber& n = g.fibers[nx];
            n.started = true;
            g.cur = &n;
            g.threadIdx = n.tid;
            continue;
        }
        break;
    }
    _longjmp(g.sched, 1);
}

inline void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& fn) {
    Global& g = G();
    const unsigned nt = block.x * block.y * block.z;
    const size_t STACK = 512 * 1024;
    if (nt > g.max_fibers) {
        if (g.stacks) munmap(g.stacks, g.stack_bytes);
        g.stack_bytes = (size_t)nt * STACK;
        g.stacks = (char*)mmap(nullptr, g.stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g.stacks == MAP_FAILED) { fprintf(stderr, "emu: mmap failed\n"); abort(); }
        g.max_fibers = nt;
    }
    if (smem_bytes > g.smem_cap) {
        free(g.smem);
        g.smem = (unsigned char*)aligned_alloc(256, (smem_bytes + 255) / 256 * 256);
        g.smem_cap = smem_bytes;
    }
    g.blockDim = block;
    g.gridDim = grid;
    g.fn = &fn;
    std::vector<Fiber> fibers(nt);
    const unsigned nwaves = (nt + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g.blockIdx = dim3(bx, by, bz);
                for (unsigned t = 0; t < nt; ++t) {
                    Fiber& f = fibers[t];
                    f.state = READY;
                    f.started = false;
                    f.flat = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                }
                g.fibers = fibers.data();
                g.nt = nt;
                g.done = 0;
                while (g.done < nt) {
                    bool ran = false;
                    for (unsigned t = 0; t < nt; ++t) {
                        Fiber& f = fibers[t];
                        if (f.state != READY) continue;
                        ran = true;
                        g.cur = &f;
                        g.threadIdx = f.tid;
                        if (_setjmp(g.sched) == 0) {
                            if (!f.started) {      // contexts are made lazily: most work-items run inside a predecessor's chain
                                f.started = true;
                                getcontext(&f.ctx);
                                f.ctx.uc_stack.ss_sp = g.stacks + (size_t)t * STACK;
                                f.ctx.uc_stack.ss_size = STACK;
                                f.ctx.uc_link = nullptr;
                                makecontext(&f.ctx, (void (*)())trampoline, 0);
                                setcontext(&f.ctx);
                            }
                            _longjmp(f.jb, 1);
                        }
                    }
                    // wave rendezvous
                    bool released = false;
                    for (unsigned w = 0; w < nwaves; ++w) {
                        unsigned lo = w * 64, hi = std::min(nt, lo + 64), waiting = 0, live = 0;
                        for (unsigned t = lo; t < hi; ++t) {
                            if (fibers[t].state != DONE) ++live;
                            if (fibers[t].state == WAIT_WAVE) ++waiting;
                        }
                        if (waiting && waiting == live) {
                            for (unsigned t = lo; t < hi; ++t)
                                if (fibers[t].state == WAIT_WAVE) fibers[t].state = READY;
                            released = true;
                        }
                    }
                    // block barrier
                    unsigned waiting = 0, live = 0;
                    for (unsigned t = 0; t < nt; ++t) {
                        if (fibers[t].state != DONE) ++live;
                        if (fibers[t].state == WAIT_BLOCK) ++waiting;
                    }
                    if (waiting && waiting == live) {
                        for (unsigned t = 0; t < nt; ++t)
                            if (fibers[t].state == WAIT_BLOCK) fibers[t].state = READY;
                        released = true;
                    }
                    if (!ran && !released && g.done < nt) {
                        fprintf(stderr, "emu: deadlock (divergent barrier) in block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g.fn = nullptr;
}

// wave exchange: every live lane of the wave must call it
template <class T>
inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "wave_exchange: <= 8 bytes");
    Global& g = G();
    unsigned flat = g.cur->flat;
    unsigned w = flat / 64;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    g.xchg[w * 64 + (flat & 63)] = raw;
    sync_wave();
    uint64_t got = g.xchg[w * 64 + (src_lane & 63)];
    sync_wave();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}

// wave ballot: every lane of the wave (none may have exited) must call it
inline unsigned long long wave_ballot(bool pred) {
    Global& g = G();
    const unsigned flat = g.cur->flat, w = flat / 64;
    const unsigned nt = g.blockDim.x * g.blockDim.y * g.blockDim.z;
    g.xchg[w * 64 + (flat & 63)] = pred ? 1 : 0;
    sync_wave();
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64 && w * 64 + l < nt; ++l)
        if (g.xchg[w * 64 + l]) m |= (unsigned long long)1 << l;
    sync_wave();
    return m;
}

}  // namespace emu

#define threadIdx (emu::G().threadIdx)
#define blockIdx (emu::G().blockIdx)
#define blockDim (emu::G().blockDim)
#define gridDim (emu::G().gridDim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __syncthreads() emu::sync_block()

static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; *p = std::max(o, v); return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
template <class T> static inline T __shfl(T v, int lane) { return emu::wave_exchange(v, lane); }
template <class T> static inline T __shfl_xor(T v, int m) { return emu::wave_exchange(v, (int)(emu::G().cur->flat & 63) ^ m); }
template <class T> static inline T __shfl_down(T v, int d) {
    int l = (int)(emu::G().cur->flat & 63);
    return emu::wave_exchange(v, l + d < 64 ? l + d : l);
}
using std::max;
using std::min;

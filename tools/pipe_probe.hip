// pipe_probe — which hardware queues of this process block one another?
// A dispatch with workgroups still waiting for a place holds the dispatcher it sits on; a kernel on another stream of the same
// dispatcher waits for the whole of it, whatever its priority.  This probe makes NS streams (the first NH of them at the highest
// priority), and for every ordered pair (i, j) launches a "hog" (four rounds of workgroups, each spinning `spin_us`) on stream i and,
// once it runs, a one-wavefront kernel on stream j; it prints the latency of the small kernel in microseconds.
//   mode 0: the hog leaves registers, LDS and wave slots free — only the queueing shows
//   mode 1: the hog takes the whole LDS of every CU and the small kernel needs LDS — it needs a place a finishing hog workgroup frees
//   mode 2: as mode 0, after streams 1 and 2 were destroyed and two more made in their stead (they take the last two places of the table):
//           does a new queue take the dispatcher of the one that went, or the next in turn?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/pipe_probe tools/pipe_probe.hip ; usage: pipe_probe [streams=12] [high=4] [mode=0] [spin_us=150]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ unsigned char dyn[];
__global__ void k_hog(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && threadIdx.x == 9999) sink[0] = dyn[0];
}
__global__ void k_small(unsigned* out) {
    if (threadIdx.x == 0) out[0] = (unsigned)wall_clock64() + dyn[0] * 0;
}
int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 12, NH = argc > 2 ? atoi(argv[2]) : 4, mode = argc > 3 ? atoi(argv[3]) : 0, spin_us = argc > 4 ? atoi(argv[4]) : 150;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::vector<hipStream_t> st(NS);
    for (int i = 0; i < NS; ++i) {
        if (i < NH) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi));
        else CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    }
    if (mode == 2) {
        CK(hipStreamDestroy(st[1]));
        CK(hipStreamDestroy(st[2]));
        st.erase(st.begin() + 1, st.begin() + 3);
        hipStream_t a, b;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        st.push_back(a);
        st.push_back(b);
        printf("# mode 2: streams made as 0 .. %d; 1 and 2 destroyed; columns now = made-order 0, 3, 4, ..., %d, then the two new ones\n", NS - 1, NS - 1);
    }
    unsigned* d;
    CK(hipMalloc(&d, 4096));
    const size_t hog_lds = mode == 1 ? 160 * 1024 : 64 * 1024, small_lds = mode == 1 ? 32 * 1024 : 0;
    CK(hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hog_lds));
    const int per_cu = mode == 1 ? 1 : 2;
    const unsigned grid = (unsigned)(cus * per_cu * 4);      // four rounds
    const unsigned long long ticks = (unsigned long long)spin_us * 100;   // the wall clock runs at 100 MHz
    printf("# %d CUs, %d streams (%d high priority first), mode %d, hog = %u workgroups x %d us (4 rounds: ~%d us), priorities %d..%d\n", cus, NS, NH, mode, grid, spin_us,
           4 * spin_us, lo, hi);
    // warm every stream
    for (int i = 0; i < NS; ++i) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), small_lds, st[i], d); CK(hipStreamSynchronize(st[i])); }
    double alone = 0;
    for (int r = 0; r < 8; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_small, dim3(1), dim3(64), small_lds, st[0], d);
        CK(hipStreamSynchronize(st[0]));
        alone += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 8;
    }
    printf("# small kernel alone: %.0f us (launch + synchronise)\n# rows: hog stream i; columns: small kernel's stream j; latency in us\n     ", alone);
    for (int j = 0; j < NS; ++j) printf("%6d", j);
    printf("\n");
    for (int i = 0; i < NS; ++i) {
        printf("%2d%s ", i, i < NH ? "H" : " ");
        for (int j = 0; j < NS; ++j) {
            if (i == j) { printf("     -"); continue; }
            hipLaunchKernelGGL(k_hog, dim3(grid), dim3(256), hog_lds, st[i], ticks, (unsigned*)nullptr);
            std::this_thread::sleep_for(std::chrono::microseconds(spin_us / 2 + 30));     // the first round is resident, three are waiting
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), small_lds, st[j], d + 1);
            CK(hipStreamSynchronize(st[j]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(st[i]));
            printf("%6.0f", us);
        }
        printf("\n");
    }
    return 0;
}

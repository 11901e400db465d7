// prequeue_latency.hip -- what a Fiat-Shamir hand-over costs between two kernels of one stream (dev tool, round 6).
// The commit phase of Fri.prove is root -> host (hash, challenge) -> next kernel.  Two ways to get the next kernel going:
//   launch     : the host sees the root in its pinned slot, THEN calls hipLaunchKernelGGL with the challenge as an argument
//                (what csrc/merkle_fri.hip does for the rounds above 2^16)
//   prequeued  : the kernel already stands in the stream behind hipStreamWaitValue64 on a word the host writes, and reads the
//                challenge from memory -- the word and the challenge in (a) pinned host memory, (b) fine-grained device memory
//                the host writes through the BAR
// Measured on the host clock: from the moment the host has seen kernel 1's flag to the moment it sees kernel 2's flag (kernel 2
// is one wave that copies the 16-byte challenge it was given to pinned memory and flags), median and minimum of 200.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/prequeue_latency tools/microbench/prequeue_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
#include <immintrin.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void first_kernel(volatile uint64_t* host, uint64_t seq) {
    if (threadIdx.x == 0) { __threadfence_system(); host[0] = seq; }
}
__global__ void second_kernel(volatile uint64_t* host, uint64_t seq, uint64_t a_lo, uint64_t a_hi, const uint64_t* challenge) {
    if (threadIdx.x == 0) {
        uint64_t lo = a_lo, hi = a_hi;
        if (challenge) { lo = __hip_atomic_load(&challenge[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); hi = __hip_atomic_load(&challenge[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        host[2] = lo; host[3] = hi;
        __threadfence_system();
        host[1] = seq;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    volatile uint64_t* host;
    CHK(hipHostMalloc((void**)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < 8; ++i) host[i] = 0;
    // (a) the challenge slot in pinned host memory, (b) in fine-grained device memory
    volatile uint64_t* slot_host;
    CHK(hipHostMalloc((void**)&slot_host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    uint64_t* slot_dev = nullptr;
    const bool have_dev = hipExtMallocWithFlags((void**)&slot_dev, 4096, hipDeviceMallocFinegrained) == hipSuccess;
    if (!have_dev) (void)hipGetLastError();
    bool dev_host_writable = false;
    if (have_dev) {
        // is device memory writable from the host (large BAR)?  Ask the runtime, do not just try
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0) == hipSuccess) dev_host_writable = large_bar != 0;
        else (void)hipGetLastError();
        CHK(hipMemset(slot_dev, 0, 4096));
    }
    printf("fine-grained device slot: %s, host-writable through the BAR: %s\n", have_dev ? "allocated" : "not available", dev_host_writable ? "yes" : "no");
    const int reps = 200;
    uint64_t seq = 0;
    auto run = [&](int mode, double& med, double& mn, bool& ok) -> int {
        std::vector<double> dt;
        ok = true;
        for (int r = 0; r < reps + 20; ++r) {
            ++seq;
            volatile uint64_t* slot = mode == 1 ? slot_host : (volatile uint64_t*)slot_dev;
            hipLaunchKernelGGL(first_kernel, dim3(1), dim3(64), 0, st, host, seq);
            if (mode != 0) {
                CHK(hipStreamWaitValue64(st, (void*)(slot + 2), seq, hipStreamWaitValueEq, ~0ull));
                hipLaunchKernelGGL(second_kernel, dim3(1), dim3(64), 0, st, host, seq, 0ull, 0ull, (const uint64_t*)slot);
            }
            while (host[0] != seq) _mm_pause();
            const double t0 = now_us();
            const uint64_t lo = seq * 0x9E3779B97F4A7C15ull, hi = ~seq;
            if (mode == 0) hipLaunchKernelGGL(second_kernel, dim3(1), dim3(64), 0, st, host, seq, lo, hi, (const uint64_t*)nullptr);
            else { slot[0] = lo; slot[1] = hi; _mm_sfence(); slot[2] = seq; _mm_sfence(); }
            double spin0 = now_us();
            while (host[1] != seq) { _mm_pause(); if (now_us() - spin0 > 2e6) { printf("  mode %d: no answer within 2 s\n", mode); return 2; } }
            const double t1 = now_us();
            if (host[2] != lo || host[3] != hi) ok = false;
            if (r >= 20) dt.push_back(t1 - t0);
            CHK(hipStreamSynchronize(st));
        }
        std::sort(dt.begin(), dt.end());
        med = dt[dt.size() / 2]; mn = dt[0];
        return 0;
    };
    double med, mn; bool ok;
    if (run(0, med, mn, ok)) return 2;
    printf("  launch after the root (challenge as kernel argument)            : median %6.2f us  min %6.2f us  challenge intact: %s\n", med, mn, ok ? "yes" : "NO");
    if (run(1, med, mn, ok) == 0)
        printf("  prequeued behind hipStreamWaitValue64, slot in pinned host memory : median %6.2f us  min %6.2f us  challenge intact: %s\n", med, mn, ok ? "yes" : "NO");
    if (have_dev && dev_host_writable) {
        if (run(2, med, mn, ok) == 0)
            printf("  prequeued, slot in fine-grained device memory (host writes BAR)  : median %6.2f us  min %6.2f us  challenge intact: %s\n", med, mn, ok ? "yes" : "NO");
    }
    return 0;
}

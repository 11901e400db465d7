// What separates the two passes of a 2^20 transform: a kernel boundary or a device-wide barrier inside one launch?  (dev tool;
// VERDICT r2 #5 proposes "one cooperative launch for both passes"; this measures the two hand-overs with the transform's own
// shape -- 256 workgroups of 1024 threads, one per CU, 64 KiB written per workgroup in pass 1 and 64 KiB read from ANOTHER
// workgroup's tile (another XCD) in pass 2 -- without touching the NTT kernels.)
//   two launches : write kernel, then read kernel (the stream orders them; the hardware writes back / invalidates the L2s)
//   one launch   : write, agent-scope release fence, arrive on a global counter, spin until all 256 have arrived, agent-scope
//                  acquire fence, read
// Both variants carry the same optional block of arithmetic per phase (SPIN iterations of dependent integer work) so that the
// hand-over is measured next to a realistic phase length.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/grid_barrier_ubench tools/microbench/grid_barrier_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WGS = 256, THREADS = 1024, PER_THREAD = 4;       // 4 x 16 bytes per thread = 64 KiB per workgroup

__device__ __forceinline__ uint4 work(uint4 v, int spin) {
    for (int i = 0; i < spin; ++i) {
        v.x = v.x * 1664525u + v.y; v.y = v.y * 22695477u + v.z; v.z = v.z * 1103515245u + v.w; v.w = v.w * 69069u + v.x;
    }
    return v;
}

__device__ __forceinline__ void write_phase(uint4* buf, uint32_t seed, int spin) {
    uint4* mine = buf + (size_t)blockIdx.x * THREADS * PER_THREAD;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        uint4 v = make_uint4(seed + blockIdx.x, threadIdx.x, k, seed);
        mine[k * THREADS + threadIdx.x] = work(v, spin);
    }
}

__device__ __forceinline__ void read_phase(const uint4* buf, uint4* out, int spin) {
    const uint32_t other = (blockIdx.x * 37u + 11u) & (WGS - 1);          // another CU, usually another XCD
    const uint4* theirs = buf + (size_t)other * THREADS * PER_THREAD;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        uint4 v = work(theirs[k * THREADS + threadIdx.x], spin);
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(THREADS) k_write(uint4* buf, uint32_t seed, int spin) { write_phase(buf, seed, spin); }
__global__ void __launch_bounds__(THREADS) k_read(const uint4* buf, uint4* out, int spin) { read_phase(buf, out, spin); }

// ONE_FENCE: every wave only waits for its own stores (workgroup-scope release), ONE thread per workgroup does the agent-scope
// release (L2 write-back), the arrival, the spin and the agent-scope acquire (L1 / L2 invalidate) -- 256 cache operations
// instead of 4096
template <bool ONE_FENCE>
__global__ void __launch_bounds__(THREADS) k_fused(uint4* buf, uint4* out, uint32_t seed, int spin, unsigned long long* counter, unsigned long long target) {
    write_phase(buf, seed, spin);
    if (ONE_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ONE_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (ONE_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!ONE_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    read_phase(buf, out, spin);
}

int main() {
    uint4 *buf, *out_a, *out_b;
    unsigned long long* counter;
    const size_t n = (size_t)WGS * THREADS * PER_THREAD;
    CHK(hipMalloc(&buf, n * sizeof(uint4)));
    CHK(hipMalloc(&out_a, (size_t)WGS * THREADS * sizeof(uint4)));
    CHK(hipMalloc(&out_b, (size_t)WGS * THREADS * sizeof(uint4)));
    CHK(hipMalloc(&counter, 8));
    CHK(hipMemset(counter, 0, 8));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    unsigned long long epoch = 0;
    printf("grid_barrier_ubench: %d workgroups x %d threads, 64 KiB written and 64 KiB read (from another workgroup) per workgroup and iteration\n", WGS, THREADS);
    for (int spin : {0, 64, 512}) {
        float best2 = 1e30f, best1 = 1e30f, best1b = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < iters; ++i) {
                hipLaunchKernelGGL(k_write, dim3(WGS), dim3(THREADS), 0, 0, buf, (uint32_t)i, spin);
                hipLaunchKernelGGL(k_read, dim3(WGS), dim3(THREADS), 0, 0, buf, out_a, spin);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best2) best2 = ms;
            hipEventRecord(e0, 0);
            for (int i = 0; i < iters; ++i) {
                epoch += WGS;
                hipLaunchKernelGGL(k_fused<false>, dim3(WGS), dim3(THREADS), 0, 0, buf, out_b, (uint32_t)i, spin, counter, epoch);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best1) best1 = ms;
            hipEventRecord(e0, 0);
            for (int i = 0; i < iters; ++i) {
                epoch += WGS;
                hipLaunchKernelGGL(k_fused<true>, dim3(WGS), dim3(THREADS), 0, 0, buf, out_b, (uint32_t)i, spin, counter, epoch);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best1b) best1b = ms;
        }
        std::vector<uint4> ha((size_t)WGS * THREADS), hb((size_t)WGS * THREADS);
        CHK(hipMemcpy(ha.data(), out_a, ha.size() * sizeof(uint4), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(hb.data(), out_b, hb.size() * sizeof(uint4), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < ha.size(); ++i) bad += ha[i].x != hb[i].x || ha[i].y != hb[i].y || ha[i].z != hb[i].z || ha[i].w != hb[i].w;
        printf("  arithmetic %4d steps per element and phase: two launches %7.2f us per iteration; one launch + device-wide barrier: fences in every wave %7.2f us (%+.2f), "
               "fences in one thread per workgroup %7.2f us (%+.2f)  results %s\n",
               spin, 1e3f * best2 / iters, 1e3f * best1 / iters, 1e3f * (best1 - best2) / iters, 1e3f * best1b / iters, 1e3f * (best1b - best2) / iters, bad ? "DIFFER" : "identical");
    }
    return 0;
}

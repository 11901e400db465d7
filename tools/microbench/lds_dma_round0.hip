// Would LDS-DMA (global_load_lds_dwordx4) for round 0 shorten a 2^20 pass?  
//
// The pass kernel at 2^20: 256 workgroups of 1024 threads, one per CU, tile = 1024 rows x 4 columns of 16-byte elements (64-byte runs
// at a 16 KiB row stride), E = 4 elements per thread.  Round 0 loads straight into registers, does its two butterfly stages there
// and writes the tile to LDS; the first barrier of the kernel comes AFTER that.  With LDS-DMA the tile lands in LDS without passing
// through registers -- but round 0's butterflies then read elements other lanes loaded, so a workgroup barrier and an LDS read pass
// come BEFORE the first butterfly.  This microbenchmark moves the tile both ways with the same trivial arithmetic:
//   A  global_load_dwordx4 x4 -> regs -> (op) -> ds_write_b128 (swizzled) -> barrier -> ds_read_b128 (other rows) -> global_store
//   B  global_load_lds_dwordx4 x4 (pre-swizzled source addresses, linear LDS) -> vmcnt(0) + barrier -> ds_read_b128 (own rows) -> (op)
//      -> ds_write_b128 -> barrier -> ds_read_b128 (other rows) -> global_store
// and, to see what the barrier alone costs,  C = A with an extra barrier + LDS round trip in front of the (op).
// hipcc --offload-arch=gfx950 -O3 -o lds_dma_round0 lds_dma_round0.hip ; ./lds_dma_round0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct __attribute__((aligned(16))) El { unsigned long long lo, hi; };
constexpr int LOGR = 10, LOGC = 2, R = 1 << LOGR, C = 1 << LOGC, THREADS = 1024, E = 4;
constexpr long N = 1L << 20;
constexpr long ROW_STRIDE = N >> LOGR;          // elements between consecutive rows of a column tile (1024)

__device__ __forceinline__ unsigned lds_index(unsigned r, unsigned c) { return (r << LOGC) | ((c ^ r) & (C - 1)); }
__device__ __forceinline__ El op(El x) { x.lo += 1; x.hi ^= x.lo; return x; }

template <int MODE>
__global__ void __launch_bounds__(THREADS) move_tile(const El* __restrict__ in, El* __restrict__ out) {
    __shared__ El lds[R * C];
    const unsigned tid = threadIdx.x, tile = blockIdx.x;
    const El* src = in + (size_t)tile * C;
    El* dst = out + (size_t)tile * C;
    El x[E];
    if (MODE == 1) {
        // one wave = 16 consecutive rows x 4 columns per load: LDS destination = wave-uniform base + lane * 16 bytes
        const unsigned lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const unsigned r = (i * 16 + wave) * 16 + (lane >> 2);            // 16 waves x 16 rows x 4 loads = 1024 rows
            const unsigned cpos = lane & 3, c = cpos ^ (r & 3);               // pre-swizzled source column, linear LDS position
            const El* g = src + (size_t)r * ROW_STRIDE + c;
            El* l = lds + ((i * 16 + wave) * 16) * C;                          // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);                                         // the DMA writes are on the VM counter
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const unsigned r = (tid >> 2) + 256 * i, c = tid & 3;
            x[i] = op(lds[lds_index(r, c)]);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const unsigned r = (tid >> 2) + 256 * i, c = tid & 3;
            x[i] = src[(size_t)r * ROW_STRIDE + c];
        }
        if (MODE == 2) {                                                       // the extra barrier + LDS round trip alone
#pragma unroll
            for (int i = 0; i < E; ++i) lds[lds_index((tid >> 2) + 256 * i, tid & 3)] = x[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < E; ++i) x[i] = lds[lds_index((tid >> 2) + 256 * i, tid & 3)];
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < E; ++i) x[i] = op(x[i]);
    }
#pragma unroll
    for (int i = 0; i < E; ++i) lds[lds_index((tid >> 2) + 256 * i, tid & 3)] = x[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; ++i) {                                              // the next round reads other rows: 4 adjacent rows per thread
        const unsigned r = (tid >> 2) * 4 + i, c = tid & 3;
        dst[(size_t)r * ROW_STRIDE + c] = lds[lds_index(r, c)];
    }
}

template <int MODE> double run(const El* in, El* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(move_tile<MODE>, dim3(256), dim3(THREADS), 0, 0, in, out);
    hipDeviceSynchronize();
    double best = 1e9;
    for (int t = 0; t < 5; ++t) {
        hipEventRecord(a, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(move_tile<MODE>, dim3(256), dim3(THREADS), 0, 0, in, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms * 1e3 / reps < best) best = ms * 1e3 / reps;
    }
    return best;
}

int main() {
    El *in, *outA, *outB, *outC;
    hipMalloc(&in, N * sizeof(El)); hipMalloc(&outA, N * sizeof(El)); hipMalloc(&outB, N * sizeof(El)); hipMalloc(&outC, N * sizeof(El));
    std::vector<El> h(N);
    for (long i = 0; i < N; ++i) h[i] = El{(unsigned long long)i * 0x9E3779B97F4A7C15ull, (unsigned long long)i};
    hipMemcpy(in, h.data(), N * sizeof(El), hipMemcpyHostToDevice);
    const double a = run<0>(in, outA, 2000), b = run<1>(in, outB, 2000), c = run<2>(in, outC, 2000);
    std::vector<El> ra(N), rb(N), rc(N);
    hipMemcpy(ra.data(), outA, N * sizeof(El), hipMemcpyDeviceToHost);
    hipMemcpy(rb.data(), outB, N * sizeof(El), hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), outC, N * sizeof(El), hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < N; ++i) bad += (ra[i].lo != rb[i].lo) || (ra[i].hi != rb[i].hi) || (ra[i].lo != rc[i].lo) || (ra[i].hi != rc[i].hi) || (ra[i].lo != h[i].lo + 1);
    printf("tile of 1024 rows x 4 columns (64 KiB) per workgroup, 256 workgroups of 1024 threads, 2^20 elements in and out (33.6 MB per launch)\n");
    printf("A  loads into registers, one barrier                          : %7.2f us per launch\n", a);
    printf("B  LDS-DMA (global_load_lds_dwordx4), barrier, LDS read first : %7.2f us per launch   (%+.1f %%)\n", b, 100 * (b - a) / a);
    printf("C  A plus the extra barrier and LDS round trip of B           : %7.2f us per launch   (%+.1f %%)\n", c, 100 * (c - a) / a);
    printf("outputs identical and correct: %s\n", bad ? "NO" : "yes");
    return bad ? 1 : 0;
}

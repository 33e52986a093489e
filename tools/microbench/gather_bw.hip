// gather_bw.hip — calibration microbenchmarks for the probe kernels' two rooflines on this box:
//   (1) streaming copy (achievable HBM bandwidth, 16 B/lane)
//   (2) random gather: out[i] = table[idx[i]] for table sizes from L2-resident to HBM-resident,
//       4- and 8-byte entries, 1..8 independent lookups in flight per thread.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_bw.hip -o gpurun_out/gather_bw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t domain) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        idx[i] = (uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * domain) >> 32);
}

template <class T, int R>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const T* __restrict__ table,
                                                uint32_t* __restrict__ out, size_t n) {
    const size_t tile = (size_t)blockIdx.x * 256 * R;
    uint32_t id[R];
    T v[R];
#pragma unroll
    for (int k = 0; k < R; k++) { size_t i = tile + (size_t)k * 256 + threadIdx.x; id[k] = i < n ? idx[i] : 0; }
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = table[id[k]];
#pragma unroll
    for (int k = 0; k < R; k++) { size_t i = tile + (size_t)k * 256 + threadIdx.x; if (i < n) out[i] = (uint32_t)v[k]; }
}

template <class T, int R>
static void run_gather(const uint32_t* idx, uint32_t* out, size_t n, size_t entries) {
    T* table;
    CK(hipMalloc(&table, entries * sizeof(T)));
    CK(hipMemset(table, 1, entries * sizeof(T)));
    uint32_t* didx = const_cast<uint32_t*>(idx);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, didx, n, (uint32_t)entries);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned grid = (unsigned)((n + 256 * R - 1) / (256 * R));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_gather<T, R>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_gather<T, R>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("gather entry=%zuB table=%8.1f MB R=%d : %7.3f ms  %6.1f Glookup/s\n", sizeof(T), entries * sizeof(T) / 1e6, R, ms,
           n / ms / 1e6);
    CK(hipFree(table));
}

int main() {
    const size_t n = 100000000;
    // (1) streaming copy, 1.6 GB in + 1.6 GB out
    {
        size_t bytes = (size_t)1600 << 20;
        uint4 *in, *out;
        CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
        CK(hipMemset(in, 1, bytes));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int grid : {2048, 8192, 65536}) {
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, in, out, bytes / 16);
            CK(hipEventRecord(a));
            for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, in, out, bytes / 16);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
            printf("copy grid=%6d: %.3f ms  %.2f TB/s (read+write)\n", grid, ms, 2.0 * bytes / ms / 1e9);
        }
        CK(hipFree(in)); CK(hipFree(out));
    }
    uint32_t *idx, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, n * 4));
    for (size_t entries : {(size_t)150000, (size_t)1000000, (size_t)10000000, (size_t)40000000, (size_t)250000000}) {
        run_gather<uint32_t, 1>(idx, out, n, entries);
        run_gather<uint32_t, 4>(idx, out, n, entries);
        run_gather<uint32_t, 8>(idx, out, n, entries);
        run_gather<uint64_t, 4>(idx, out, n, entries);
        run_gather<uint64_t, 8>(idx, out, n, entries);
    }
    return 0;
}

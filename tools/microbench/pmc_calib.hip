// pmc_calib.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this box against KNOWN byte counts, for
// exactly the access classes of the chained-join kernel (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports 1/2 of a
// wide coalesced streaming read ... other access widths are uncalibrated: calibrate in your own access pattern"):
//   read_b4 / read_b8 / read_b16   coalesced streaming reads of 4 / 8 / 16 bytes per lane   (1 GiB each)
//   write_b4 / write_b16           coalesced streaming writes                               (1 GiB each)
//   gather_40MB / gather_4GB       1e8 random 4-byte lookups (Infinity-Cache sized / HBM sized table)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/gpu_pmc_calib.sh); the
// kernel names carry the expected bytes.  Buffers are > 256 MiB so the Infinity Cache cannot hide re-reads.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T>
__global__ __launch_bounds__(256) void read_stream(const T* __restrict__ in, size_t n, uint32_t* __restrict__ sink) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T v = in[i];
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void write_stream(T* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    T v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = threadIdx.x + k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
// indices are computed, not read: the only memory traffic is the gather itself (+ a 4-byte result per row)
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ table, uint32_t entries, size_t n,
                                              uint32_t* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = table[(uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * entries) >> 32)];
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *a, *b;
    uint32_t* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, (size_t)4 << 30)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 1, (size_t)4 << 30));
    CK(hipDeviceSynchronize());
    const dim3 g(8192), t(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read_stream<uint32_t>, g, t, 0, 0, (const uint32_t*)a, bytes / 4, sink);
        hipLaunchKernelGGL(read_stream<uint2>, g, t, 0, 0, (const uint2*)a, bytes / 8, sink);
        hipLaunchKernelGGL(read_stream<uint4>, g, t, 0, 0, (const uint4*)a, bytes / 16, sink);
        hipLaunchKernelGGL(write_stream<uint32_t>, g, t, 0, 0, (uint32_t*)a, bytes / 4);
        hipLaunchKernelGGL(write_stream<uint4>, g, t, 0, 0, (uint4*)a, bytes / 16);
        hipLaunchKernelGGL(gather, g, t, 0, 0, (const uint32_t*)b, 10000000u, (size_t)100000000, (uint32_t*)a);
        hipLaunchKernelGGL(gather, g, t, 0, 0, (const uint32_t*)b, 1000000000u, (size_t)100000000, (uint32_t*)a);
        CK(hipDeviceSynchronize());
    }
    printf("expected bytes per launch: read_stream<*> %zu read / 0 written; write_stream<*> 0 / %zu; "
           "gather(40 MB table) 1e8 sectors x 64 B = 6.4e9 read (less what the caches absorb) / 4e8 written; "
           "gather(4 GB table) 6.4e9 read / 4e8 written\n", bytes, bytes);
    return 0;
}

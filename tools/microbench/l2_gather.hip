// l2_gather.hip — what one random lookup into an L2-resident table costs on gfx950, by load width and cache policy
// (round 5: the rank-table lookups of k_chain_dense are bound by the L2 -> L1 path; does any policy bit or entry width move it?)
//   out[i] = f(table[idx[i]]) for 1e8 random idx, tables of 0.6 .. 5 MB, entries of 4 / 8 / 16 bytes, 8 lookups in flight per lane,
//   policies: default | sc0 | sc1 | sc0 sc1 | nt | nt sc0 sc1  (global_load_* ... off <bits>)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_gather.hip -o gpurun_out/l2_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t domain) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        idx[i] = (uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * domain) >> 32);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LOADS(NAME, BITS)                                                                                                   \
    __device__ __forceinline__ void NAME##_b32(uint32_t& v, const void* p) { asm volatile("global_load_dword %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory"); }   \
    __device__ __forceinline__ void NAME##_b64(u32x2& v, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory"); }   \
    __device__ __forceinline__ void NAME##_b128(u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory"); }
LOADS(p0, "")
LOADS(p1, "sc0")
LOADS(p2, "sc1")
LOADS(p3, "sc0 sc1")
LOADS(p4, "nt")
LOADS(p5, "sc0 sc1 nt")

constexpr int R = 8;
template <int W, int P>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ table, uint32_t* __restrict__ out, size_t n) {
    const size_t tile = (size_t)blockIdx.x * 256 * R;
    uint32_t id[R];
#pragma unroll
    for (int k = 0; k < R; k++) { size_t i = tile + (size_t)k * 256 + threadIdx.x; id[k] = i < n ? idx[i] : 0; }
    uint32_t a[R];
    u32x2 b[R];
    u32x4 c[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint8_t* p = table + (size_t)id[k] * W;
        if constexpr (W == 4) {
            if constexpr (P == 0) p0_b32(a[k], p); else if constexpr (P == 1) p1_b32(a[k], p); else if constexpr (P == 2) p2_b32(a[k], p);
            else if constexpr (P == 3) p3_b32(a[k], p); else if constexpr (P == 4) p4_b32(a[k], p); else p5_b32(a[k], p);
        } else if constexpr (W == 8) {
            if constexpr (P == 0) p0_b64(b[k], p); else if constexpr (P == 1) p1_b64(b[k], p); else if constexpr (P == 2) p2_b64(b[k], p);
            else if constexpr (P == 3) p3_b64(b[k], p); else if constexpr (P == 4) p4_b64(b[k], p); else p5_b64(b[k], p);
        } else {
            if constexpr (P == 0) p0_b128(c[k], p); else if constexpr (P == 1) p1_b128(c[k], p); else if constexpr (P == 2) p2_b128(c[k], p);
            else if constexpr (P == 3) p3_b128(c[k], p); else if constexpr (P == 4) p4_b128(c[k], p); else p5_b128(c[k], p);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < R; k++) {
        size_t i = tile + (size_t)k * 256 + threadIdx.x;
        uint32_t v;
        if constexpr (W == 4) v = a[k]; else if constexpr (W == 8) v = b[k].x + b[k].y; else v = c[k].x + c[k].y + c[k].z + c[k].w;
        if (i < n) __builtin_nontemporal_store(v, out + i);
    }
}

template <int W, int P>
static void run(const uint32_t* idx, uint32_t* out, size_t n, size_t entries, const uint8_t* table) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned grid = (unsigned)((n + 256 * R - 1) / (256 * R));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_gather<W, P>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_gather<W, P>), dim3(grid), dim3(256), 0, 0, idx, table, out, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    static const char* names[] = {"default", "sc0", "sc1", "sc0 sc1", "nt", "sc0 sc1 nt"};
    printf("entry=%2dB table=%6.2f MB policy=%-10s : %7.3f ms  %6.1f Glookup/s\n", W, entries * (double)W / 1e6, names[P], ms, n / ms / 1e6);
}

template <int W>
static void sweep(uint32_t* idx, uint32_t* out, size_t n, size_t table_bytes) {
    const size_t entries = table_bytes / W;
    uint8_t* table;
    CK(hipMalloc(&table, table_bytes));
    CK(hipMemset(table, 1, table_bytes));
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, n, (uint32_t)entries);
    run<W, 0>(idx, out, n, entries, table);
    run<W, 1>(idx, out, n, entries, table);
    run<W, 2>(idx, out, n, entries, table);
    run<W, 3>(idx, out, n, entries, table);
    run<W, 4>(idx, out, n, entries, table);
    run<W, 5>(idx, out, n, entries, table);
    CK(hipFree(table));
}

int main() {
    const size_t n = 100000000;
    uint32_t *idx, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, n * 4));
    for (size_t bytes : {(size_t)640 << 10, (size_t)1280 << 10, (size_t)2560 << 10, (size_t)5120 << 10, (size_t)40 << 20}) {
        sweep<4>(idx, out, n, bytes);
        sweep<8>(idx, out, n, bytes);
        sweep<16>(idx, out, n, bytes);
    }
    return 0;
}

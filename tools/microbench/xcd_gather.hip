// xcd_gather.hip — experiment: can 1e8 random lookups into a 40 MB table beat the ~55 G lookups/s
// ceiling of plain gathers (profiles/r01_gather_copy_calibration.txt) by making the table L2-resident?
//
// Every XCD has its own 4 MiB L2.  The table is cut into P slices by code range; workgroup b serves
// slice (b % 8) [+8, +16 ... when P > 8] — observed placement: block b runs on XCD b % 8 — and EVERY
// slice owner streams ALL the codes (the 7 re-reads are expected to hit the Infinity Cache), keeps
// the rows whose code falls into its slice, looks them up (slice = T*4/P bytes, L2-sized) and writes
// the results compactly per (1024-row chunk, slice) with the row's offset inside the chunk.  A merge
// pass then re-interleaves the P compact segments of a chunk through LDS: no random HBM access at all.
//
//   pass A proxy   read 8 B/row, write a 4-byte code per row
//   pass B         the slice-owner pass above                       <- what this file is about
//   pass C proxy   merge: chunk's segments -> LDS -> out[row]
//   baseline       out[i] = table[codes[i]]  (plain gather, 4 rows in flight per thread)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/xcd_gather.hip -o gpurun_out/xcd_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChunk = 1024;            // rows per wave-chunk
constexpr uint32_t kAbsent = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__global__ void k_fill_codes(uint32_t* codes, uint64_t* keys, size_t n, uint32_t domain) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        codes[i] = (uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * domain) >> 32);
        keys[i] = codes[i];
    }
}
__global__ void k_fill_table(uint32_t* t, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) t[i] = mix((uint32_t)i) & 0x00FFFFFFu;
}
__global__ void k_check(const uint32_t* codes, const uint32_t* table, const uint32_t* out, size_t n, unsigned long long* bad) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b += out[i] != table[codes[i]];
    if (b) atomicAdd(bad, b);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t wave_incl(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d, 64); if (lane_id() >= d) v += o; }
    return v;
}

// ---- baseline -----------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ table,
                                                uint32_t* __restrict__ out, size_t n) {
    const size_t tile = (size_t)blockIdx.x * 256 * R;
    uint32_t id[R], v[R];
#pragma unroll
    for (int k = 0; k < R; k++) { size_t i = tile + (size_t)k * 256 + threadIdx.x; id[k] = i < n ? idx[i] : 0; }
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = table[id[k]];
#pragma unroll
    for (int k = 0; k < R; k++) { size_t i = tile + (size_t)k * 256 + threadIdx.x; if (i < n) out[i] = v[k]; }
}

// ---- pass A proxy --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_passA(const ulonglong2* __restrict__ keys, uint2* __restrict__ codes, size_t npairs) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
        ulonglong2 k = keys[i];
        codes[i] = make_uint2((uint32_t)k.x, (uint32_t)k.y);
    }
}

// ---- pass B ----------------------------------------------------------------------------------------------
// NT: non-temporal loads of the code stream.  LOOKUP=false: attribution (stream + filter + compact only).
template <bool NT, bool LOOKUP>
__global__ __launch_bounds__(256) void k_passB(const uint32_t* __restrict__ codes, uint32_t nchunks,
                                               const uint32_t* __restrict__ table, uint32_t T, int P,
                                               uint32_t* __restrict__ seg, uint16_t* __restrict__ seg_off,
                                               uint32_t* __restrict__ counts, uint32_t* __restrict__ xcc_hist) {
    __shared__ uint32_t q_code[4][kChunk];
    __shared__ uint16_t q_off[4][kChunk];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t G = gridDim.x >> 3, local = blockIdx.x >> 3;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(6164) & 15u;   // HW_REG_XCC_ID[3:0]
        atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xcc], 1u);
    }
    for (int part = blockIdx.x & 7; part < P; part += 8) {
        const uint32_t lo = (uint32_t)((uint64_t)part * T / P), hi = (uint32_t)((uint64_t)(part + 1) * T / P);
        for (uint32_t tile = local; tile * 4 < nchunks; tile += G) {
            const uint32_t chunk = tile * 4 + wave;
            if (chunk >= nchunks) continue;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4* src = reinterpret_cast<const u32x4*>(codes + (size_t)chunk * kChunk);
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = NT ? __builtin_nontemporal_load(&src[j * 64 + lane]) : src[j * 64 + lane];
            uint32_t running = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                bool m[4];
                uint32_t cnt = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { m[k] = c[k] >= lo && c[k] < hi; cnt += m[k]; }
                const uint32_t incl = wave_incl(cnt);
                uint32_t pos = running + incl - cnt;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (m[k]) { q_code[wave][pos] = c[k]; q_off[wave][pos] = (uint16_t)(j * 256 + lane * 4 + k); pos++; }
                running += __shfl(incl, 63, 64);
            }
            __builtin_amdgcn_wave_barrier();
            const size_t base = ((size_t)chunk * P + part) * kChunk;
            for (uint32_t i0 = 0; i0 < running; i0 += 256) {
                uint32_t cc[4], e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint32_t i = i0 + k * 64 + lane; cc[k] = i < running ? q_code[wave][i] : lo; }
#pragma unroll
                for (int k = 0; k < 4; k++) e[k] = LOOKUP ? table[cc[k]] : cc[k];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = i0 + k * 64 + lane;
                    if (i < running) { seg[base + i] = e[k]; seg_off[base + i] = q_off[wave][i]; }
                }
            }
            if (lane == 0) counts[(size_t)chunk * P + part] = running;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- pass C proxy ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_passC(const uint32_t* __restrict__ seg, const uint16_t* __restrict__ seg_off,
                                               const uint32_t* __restrict__ counts, uint32_t nchunks, int P,
                                               uint32_t* __restrict__ out) {
    __shared__ uint32_t s_out[4][kChunk];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile * 4 < nchunks; tile += gridDim.x) {
        const uint32_t chunk = tile * 4 + wave;
        if (chunk >= nchunks) continue;
#pragma unroll
        for (int j = 0; j < 16; j++) s_out[wave][j * 64 + lane] = kAbsent;
        __builtin_amdgcn_wave_barrier();
        uint32_t n = lane < P ? counts[(size_t)chunk * P + lane] : 0;
        for (int p = 0; p < P; p++) {
            const uint32_t np = __shfl(n, p, 64);
            const size_t base = ((size_t)chunk * P + p) * kChunk;
            for (uint32_t i = lane; i < np; i += 64) s_out[wave][seg_off[base + i]] = seg[base + i];
        }
        __builtin_amdgcn_wave_barrier();
        uint4* dst = reinterpret_cast<uint4*>(out + (size_t)chunk * kChunk);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t* s = &s_out[wave][j * 256 + lane * 4];
            dst[j * 64 + lane] = make_uint4(s[0], s[1], s[2], s[3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// ======================================================================================================
// Second design: partition once instead of letting every slice owner re-read the stream.
//   A2  per 1024-row chunk: p = slice of the code; the chunk's rows are multi-split through LDS into
//       staged[(chunk*P + p)*CAP + i] = (code - lo_p) << 10 | row offset in the chunk   (any order inside a
//       segment: the offset travels with the entry), counts[chunk*P + p].  CAP = 2x the expected count; rows
//       past CAP are dropped here and resolved by a plain lookup in C2 (skewed keys).
//   B2  workgroups of XCD x serve the slices p = x, x+8, ...: res[...] = table[lo_p + (w >> 10)]
//   C2  merge: out[chunk*1024 + (w & 1023)] = res   via LDS
template <int CAP>
__global__ __launch_bounds__(256) void k_passA2(const uint32_t* __restrict__ codes, uint32_t nchunks, uint32_t T, int P,
                                                uint32_t pscale, const uint32_t* __restrict__ lo_of,
                                                uint32_t* __restrict__ staged, uint32_t* __restrict__ counts,
                                                size_t cs, size_t ps) {
    extern __shared__ uint32_t smem2[];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t* s_cnt = smem2 + wave * (64 + P * CAP);      // [64] counters (P <= 64), then [P][CAP] staged words
    uint32_t* s_stage = s_cnt + 64;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t tile = blockIdx.x; tile * 4 < nchunks; tile += gridDim.x) {
        const uint32_t chunk = tile * 4 + wave;
        if (chunk >= nchunks) continue;
        s_cnt[lane] = 0;
        const u32x4* src = reinterpret_cast<const u32x4*>(codes + (size_t)chunk * kChunk);
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = src[j * 64 + lane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t c[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t p = __umulhi(c[k], pscale);
                p = p < (uint32_t)P ? p : (uint32_t)P - 1;
                const uint32_t pos = atomicAdd(&s_cnt[p], 1u);
                if (pos < CAP) s_stage[p * CAP + pos] = ((c[k] - lo_of[p]) << 10) | (uint32_t)(j * 256 + lane * 4 + k);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const size_t base = (size_t)chunk * cs;
        for (int p = 0; p < P; p++) {
            const uint32_t np = min(s_cnt[p], (uint32_t)CAP);
            for (uint32_t i = lane; i < np; i += 64) staged[base + p * ps + i] = s_stage[p * CAP + i];
        }
        if (lane < P) {
            counts[(size_t)chunk * P + lane] = s_cnt[lane];                              // chunk-major (for C)
            counts[(size_t)nchunks * P + (size_t)lane * nchunks + chunk] = s_cnt[lane];  // slice-major (for B)
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int CAP, int U>
__global__ __launch_bounds__(256) void k_passB2(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ counts,
                                                uint32_t nchunks, const uint32_t* __restrict__ table, int P,
                                                const uint32_t* __restrict__ lo_of, uint32_t* __restrict__ res,
                                                size_t cs, size_t ps) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t G = gridDim.x >> 3, local = blockIdx.x >> 3;
    const uint32_t* countsT = counts + (size_t)nchunks * P;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(local * 4 + wave), nw = G * 4;
    for (int p = blockIdx.x & 7; p < P; p += 8) {
        const uint32_t lo = lo_of[p];
        for (uint32_t c0 = wid * U; c0 < nchunks; c0 += nw * U) {
            uint32_t n[U], w[U][CAP / 64], e[U][CAP / 64];
#pragma unroll
            for (int u = 0; u < U; u++) n[u] = c0 + u < nchunks ? min(countsT[(size_t)p * nchunks + c0 + u], (uint32_t)CAP) : 0u;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < CAP / 64; h++) {
                    const uint32_t i = h * 64 + lane;
                    w[u][h] = i < n[u] ? staged[(size_t)(c0 + u) * cs + p * ps + i] : 0u;
                }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < CAP / 64; h++) {
                    const uint32_t i = h * 64 + lane;
                    e[u][h] = i < n[u] ? table[lo + (w[u][h] >> 10)] : 0u;
                }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < CAP / 64; h++) {
                    const uint32_t i = h * 64 + lane;
                    if (i < n[u]) res[(size_t)(c0 + u) * cs + p * ps + i] = e[u][h];
                }
        }
    }
}


// B3: like B2 but the U segments of an iteration are concatenated so that every lookup instruction has 64 busy
// lanes (a segment holds ~CAP/2 entries), and the table load flavour is selectable:
//   MODE 0 plain, 1 non-temporal, 2 relaxed agent-scope atomic load (sc1: bypasses the CU's L1)
template <int MODE>
__device__ __forceinline__ uint32_t table_load(const uint32_t* p) {
    if constexpr (MODE == 1) return __builtin_nontemporal_load(p);
    else if constexpr (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <int CAP, int U, int MODE>
__global__ __launch_bounds__(256) void k_passB3(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ counts,
                                                uint32_t nchunks, const uint32_t* __restrict__ table, int P,
                                                const uint32_t* __restrict__ lo_of, uint32_t* __restrict__ res,
                                                size_t cs, size_t ps) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t G = gridDim.x >> 3, local = blockIdx.x >> 3;
    const uint32_t* countsT = counts + (size_t)nchunks * P;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(local * 4 + wave), nw = G * 4;
    constexpr int K = U * CAP / 64 / 2 + 1;      // lookup instructions per iteration when segments are ~half full
    for (int p = blockIdx.x & 7; p < P; p += 8) {
        const uint32_t lo = lo_of[p];
        for (uint32_t c0 = wid * U; c0 < nchunks; c0 += nw * U) {
            uint32_t pre[U + 1];
            pre[0] = 0;
#pragma unroll
            for (int u = 0; u < U; u++)
                pre[u + 1] = pre[u] + (c0 + u < nchunks ? min(countsT[(size_t)p * nchunks + c0 + u], (uint32_t)CAP) : 0u);
            const uint32_t total = pre[U];
            for (uint32_t f0 = 0; f0 < total; f0 += K * 64) {
                size_t addr[K];
                uint32_t w[K], e[K];
                bool on[K];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const uint32_t f = f0 + k * 64 + lane;
                    on[k] = f < total;
                    uint32_t u = 0;
#pragma unroll
                    for (int t = 1; t < U; t++) u += f >= pre[t];
                    uint32_t pu = 0;
#pragma unroll
                    for (int t = 1; t < U; t++) pu = f >= pre[t] ? pre[t] : pu;
                    addr[k] = (size_t)(c0 + u) * cs + p * ps + (f - pu);
                    w[k] = on[k] ? staged[addr[k]] : 0u;
                }
#pragma unroll
                for (int k = 0; k < K; k++) e[k] = on[k] ? table_load<MODE>(&table[lo + (w[k] >> 10)]) : 0u;
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (on[k]) res[addr[k]] = e[k];
            }
        }
    }
}

// calibration: what a perfectly dense gather from XCD-local slices does (no segments, no counts): workgroups of
// XCD x read idx from their own eighth of the rows and look up in slice x of the table only
template <int MODE>
__global__ __launch_bounds__(256) void k_local_gather(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ table,
                                                      uint32_t* __restrict__ out, size_t n, uint32_t slice) {
    const uint32_t x = blockIdx.x & 7, G = gridDim.x >> 3, local = blockIdx.x >> 3;
    const size_t per = n / 8, begin = per * x;
    for (size_t t = (size_t)local * 1024; t < per; t += (size_t)G * 1024) {
        uint32_t id[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { size_t i = t + k * 256 + threadIdx.x; id[k] = i < per ? idx[begin + i] % slice : 0; }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = table_load<MODE>(&table[(size_t)x * slice + id[k]]);
#pragma unroll
        for (int k = 0; k < 4; k++) { size_t i = t + k * 256 + threadIdx.x; if (i < per) out[begin + i] = v[k]; }
    }
}

template <int CAP>
__global__ __launch_bounds__(256) void k_passC2(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ res,
                                                const uint32_t* __restrict__ counts, uint32_t nchunks, int P,
                                                uint32_t* __restrict__ out, size_t cs, size_t ps) {
    __shared__ uint32_t s_out[4][kChunk];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t tile = blockIdx.x; tile * 4 < nchunks; tile += gridDim.x) {
        const uint32_t chunk = tile * 4 + wave;
        if (chunk >= nchunks) continue;
#pragma unroll
        for (int j = 0; j < 16; j++) s_out[wave][j * 64 + lane] = kAbsent;
        __builtin_amdgcn_wave_barrier();
        const uint32_t nl = lane < P ? min(counts[(size_t)chunk * P + lane], (uint32_t)CAP) : 0;
        const size_t base = (size_t)chunk * cs;
        for (int p0 = 0; p0 < P; p0 += 4) {
            uint32_t w[4][CAP / 64], e[4][CAP / 64], np[4];
#pragma unroll
            for (int u = 0; u < 4; u++) np[u] = __shfl(nl, p0 + u, 64);
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int h = 0; h < CAP / 64; h++) {
                    const uint32_t i = h * 64 + lane;
                    const bool on = i < np[u];
                    w[u][h] = on ? staged[base + (p0 + u) * ps + i] : 0u;
                    e[u][h] = on ? res[base + (p0 + u) * ps + i] : 0u;
                }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int h = 0; h < CAP / 64; h++)
                    if ((uint32_t)(h * 64 + lane) < np[u]) s_out[wave][w[u][h] & 1023u] = e[u][h];
        }
        __builtin_amdgcn_wave_barrier();
        uint4* dst = reinterpret_cast<uint4*>(out + (size_t)chunk * kChunk);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t* s = &s_out[wave][j * 256 + lane * 4];
            dst[j * 64 + lane] = make_uint4(s[0], s[1], s[2], s[3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> float run(F f, int reps = 5, int warm = 2) {
        for (int i = 0; i < warm; i++) f();
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; i++) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        return ms / reps;
    }
};

int main(int argc, char** argv) {
    const uint32_t T = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    const uint32_t nchunks = argc > 2 ? (uint32_t)atol(argv[2]) : 97657u;
    const int maxP = argc > 3 ? atoi(argv[3]) : 32;
    const size_t n = (size_t)nchunks * kChunk;
    printf("rows=%zu table=%u entries (%.1f MB)\n", n, T, T * 4 / 1e6);
    uint32_t *codes, *table, *out, *counts, *seg, *xcc;
    uint64_t* keys;
    uint16_t* seg_off;
    unsigned long long* bad;
    CK(hipMalloc(&codes, n * 4)); CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&table, (size_t)T * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMalloc(&counts, (size_t)nchunks * maxP * 4 * 2));
    CK(hipMalloc(&seg, n * maxP * 4 + (64u << 20))); CK(hipMalloc(&seg_off, n * maxP * 2));
    CK(hipMalloc(&xcc, 8 * 16 * 4)); CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(k_fill_codes, dim3(4096), dim3(256), 0, 0, codes, keys, n, T);
    hipLaunchKernelGGL(k_fill_table, dim3(4096), dim3(256), 0, 0, table, (size_t)T);
    CK(hipDeviceSynchronize());
    Timer tm;
    auto check = [&](const char* what) {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, codes, table, out, n, bad);
        unsigned long long h; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("  check %s: %llu mismatches\n", what, h);
    };
    {
        unsigned grid = (unsigned)((n + 1023) / 1024);
        float ms = tm.run([&] { hipLaunchKernelGGL((k_gather<4>), dim3(grid), dim3(256), 0, 0, codes, table, out, n); });
        printf("baseline plain gather R=4          : %7.3f ms  %6.1f Glookup/s\n", ms, n / ms / 1e6);
        check("baseline");
    }
    {
        float ms = tm.run([&] { hipLaunchKernelGGL(k_passA, dim3(8192), dim3(256), 0, 0, (const ulonglong2*)keys, (uint2*)out, n / 2); });
        printf("pass A proxy (8 B in, 4 B out /row): %7.3f ms  %6.2f TB/s\n", ms, n * 12.0 / ms / 1e9);
    }
    for (int P : {8, 16, 32}) {
        if (P > maxP) break;
        for (int blocks_per_cu : {4, 6}) {
            const unsigned grid = 256u * blocks_per_cu;
            for (int variant = 0; variant < 3; variant++) {   // 0: plain loads, 1: nt stream, 2: no lookup (attribution)
                CK(hipMemset(xcc, 0, 8 * 16 * 4));
                auto launch = [&] {
                    if (variant == 0) hipLaunchKernelGGL((k_passB<false, true>), dim3(grid), dim3(256), 0, 0, codes, nchunks, table, T, P, seg, seg_off, counts, xcc);
                    if (variant == 1) hipLaunchKernelGGL((k_passB<true, true>), dim3(grid), dim3(256), 0, 0, codes, nchunks, table, T, P, seg, seg_off, counts, xcc);
                    if (variant == 2) hipLaunchKernelGGL((k_passB<false, false>), dim3(grid), dim3(256), 0, 0, codes, nchunks, table, T, P, seg, seg_off, counts, xcc);
                };
                float ms = tm.run(launch, 4, 1);
                printf("pass B P=%2d slice=%5.2f MB wg/CU=%d %-9s: %7.3f ms  %6.1f Glookup/s\n", P, T * 4.0 / P / 1e6, blocks_per_cu,
                       variant == 0 ? "plain" : variant == 1 ? "nt-stream" : "no-lookup", ms, n / ms / 1e6);
            }
        }
        // leave the lookup variant's output in place for pass C
        hipLaunchKernelGGL((k_passB<false, true>), dim3(1536), dim3(256), 0, 0, codes, nchunks, table, T, P, seg, seg_off, counts, xcc);
        float ms = tm.run([&] { hipLaunchKernelGGL(k_passC, dim3(2048), dim3(256), 0, 0, seg, seg_off, counts, nchunks, P, out); });
        printf("pass C proxy (merge) P=%2d          : %7.3f ms\n", P, ms);
        check("B+C");
    }

    // ---- second design ---------------------------------------------------------------------------------
    {
        uint32_t* d_lo;
        CK(hipMalloc(&d_lo, 65 * 4));
        auto run2 = [&](auto capc, int P, int layout) {
            constexpr int CAP = decltype(capc)::value;
            if (P * CAP > 4096 || P * CAP < 1536) return;
            const uint32_t pscale = (uint32_t)((((uint64_t)P << 32) + T - 1) / T);
            std::vector<uint32_t> lo(65, 0);
            for (int p = 0; p <= P; p++) lo[p] = (uint32_t)((((uint64_t)p << 32) + pscale - 1) / pscale);
            lo[0] = 0;
            CK(hipMemcpy(d_lo, lo.data(), 65 * 4, hipMemcpyHostToDevice));
            // layout 0: chunk-major, dense (stride P*CAP words: a power of two);  1: chunk-major + 64 words of padding;
            //        2: slice-major (each slice's segments contiguous), slice stride padded by 320 words
            size_t cs, ps;
            if (layout == 0) { cs = (size_t)P * CAP; ps = CAP; }
            else if (layout == 1) { cs = (size_t)P * CAP + 64; ps = CAP; }
            else { cs = CAP; ps = (size_t)nchunks * CAP + 320; }
            const size_t words = layout == 2 ? ps * P : cs * nchunks;
            uint32_t* staged = seg;                       // reuse the big buffers
            uint32_t* res = seg + words;
            const size_t ldsA = 4 * (64 + P * CAP) * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_passA2<CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
            float a = tm.run([&] { hipLaunchKernelGGL((k_passA2<CAP>), dim3(2048), dim3(256), ldsA, 0, codes, nchunks, T, P, pscale, d_lo, staged, counts, cs, ps); });
            float b2 = tm.run([&] { hipLaunchKernelGGL((k_passB2<CAP, 2>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            float b4 = tm.run([&] { hipLaunchKernelGGL((k_passB2<CAP, 4>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            float b8 = tm.run([&] { hipLaunchKernelGGL((k_passB2<CAP, 8>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            float b3[3][2];
            b3[0][0] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 4, 0>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            b3[1][0] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 4, 1>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            b3[2][0] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 4, 2>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            b3[0][1] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 8, 0>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            b3[1][1] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 8, 1>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            b3[2][1] = tm.run([&] { hipLaunchKernelGGL((k_passB3<CAP, 8, 2>), dim3(2048), dim3(256), 0, 0, staged, counts, nchunks, table, P, d_lo, res, cs, ps); });
            printf("   B3 flattened: U4 plain %6.3f nt %6.3f sc1 %6.3f | U8 plain %6.3f nt %6.3f sc1 %6.3f\n", b3[0][0], b3[1][0], b3[2][0], b3[0][1], b3[1][1], b3[2][1]);
            float c = tm.run([&] { hipLaunchKernelGGL((k_passC2<CAP>), dim3(2048), dim3(256), 0, 0, staged, res, counts, nchunks, P, out, cs, ps); });
            float bb = b2 < b4 ? b2 : b4; bb = bb < b8 ? bb : b8;
            printf("design 2 layout %d P=%2d CAP=%3d slice=%5.2f MB: A2 %6.3f  B2(U2) %6.3f  B2(U4) %6.3f  B2(U8) %6.3f  C2 %6.3f ms   A2+B2+C2 = %6.3f ms\n",
                   layout, P, CAP, T * 4.0 / P / 1e6, a, b2, b4, b8, c, a + bb + c);
            check("A2+B2+C2");
        };
        for (uint32_t slice : {312500u, 625000u, 1250000u}) {
            float t0 = tm.run([&] { hipLaunchKernelGGL((k_local_gather<0>), dim3(2048), dim3(256), 0, 0, codes, table, out, n, slice); });
            float t1 = tm.run([&] { hipLaunchKernelGGL((k_local_gather<1>), dim3(2048), dim3(256), 0, 0, codes, table, out, n, slice); });
            float t2 = tm.run([&] { hipLaunchKernelGGL((k_local_gather<2>), dim3(2048), dim3(256), 0, 0, codes, table, out, n, slice); });
            printf("XCD-local dense gather, slice %5.2f MB: plain %6.3f  nt %6.3f  sc1 %6.3f ms\n", slice * 4 / 1e6, t0, t1, t2);
        }
        for (int layout = 1; layout < 2; layout++)
        for (int P : {8, 16, 24, 32}) {
            run2(std::integral_constant<int, 64>{}, P, layout);
            run2(std::integral_constant<int, 128>{}, P, layout);
            run2(std::integral_constant<int, 256>{}, P, layout);
        }
    }

    // ---- design 2, passes overlapped: row blocks pipelined over three streams --------------------------------
    // A(b) -> B(b) -> C(b) per row block b; A / C are HBM-bound, B is bound by the L2->L1 path: do they overlap?
    {
        constexpr int CAP = 128;
        const int P = 16;
        const uint32_t pscale = (uint32_t)((((uint64_t)P << 32) + T - 1) / T);
        std::vector<uint32_t> lo(65, 0);
        for (int p = 0; p <= P; p++) lo[p] = (uint32_t)((((uint64_t)p << 32) + pscale - 1) / pscale);
        lo[0] = 0;
        uint32_t* d_lo2;
        CK(hipMalloc(&d_lo2, 65 * 4));
        CK(hipMemcpy(d_lo2, lo.data(), 65 * 4, hipMemcpyHostToDevice));
        const size_t cs = (size_t)P * CAP + 64, ps = CAP;
        const size_t ldsA = 4 * (64 + P * CAP) * 4;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_passA2<CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
        hipStream_t sA, sB, sC;
        CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
        for (int K : {1, 2, 4, 8}) {
            for (unsigned gAC : {2048u, 1024u, 512u}) {
                for (unsigned gB : {2048u, 1024u}) {
                    const uint32_t per = (nchunks + K - 1) / K;
                    std::vector<hipEvent_t> eA(K), eB(K);
                    for (int b = 0; b < K; b++) { CK(hipEventCreateWithFlags(&eA[b], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB[b], hipEventDisableTiming)); }
                    auto run_once = [&](bool overlapped) {
                        for (int b = 0; b < K; b++) {
                            const uint32_t c0 = b * per, nb = c0 + per <= nchunks ? per : nchunks - c0;
                            uint32_t* staged = seg + (size_t)c0 * cs;
                            uint32_t* res = seg + (size_t)nchunks * cs + (size_t)c0 * cs;
                            uint32_t* cnt = counts + (size_t)c0 * P * 2;   // this block's chunk-major + slice-major counts
                            hipStream_t a = overlapped ? sA : sA, bs = overlapped ? sB : sA, c = overlapped ? sC : sA;
                            hipLaunchKernelGGL((k_passA2<CAP>), dim3(gAC), dim3(256), ldsA, a, codes + (size_t)c0 * kChunk, nb, T, P, pscale, d_lo2, staged, cnt, cs, ps);
                            if (overlapped) { CK(hipEventRecord(eA[b], a)); CK(hipStreamWaitEvent(bs, eA[b], 0)); }
                            hipLaunchKernelGGL((k_passB2<CAP, 4>), dim3(gB), dim3(256), 0, bs, staged, cnt, nb, table, P, d_lo2, res, cs, ps);
                            if (overlapped) { CK(hipEventRecord(eB[b], bs)); CK(hipStreamWaitEvent(c, eB[b], 0)); }
                            hipLaunchKernelGGL((k_passC2<CAP>), dim3(gAC), dim3(256), 0, c, staged, res, cnt, nb, P, out + (size_t)c0 * kChunk, cs, ps);
                        }
                    };
                    float t[2];
                    for (int ov = 0; ov < 2; ov++) {
                        run_once(ov); CK(hipDeviceSynchronize());
                        auto h0 = std::chrono::steady_clock::now();
                        for (int r = 0; r < 5; r++) run_once(ov);
                        CK(hipDeviceSynchronize());
                        t[ov] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - h0).count() / 5;
                    }
                    printf("overlap  K=%d row blocks, grids A/C %4u B %4u : serial %6.3f ms  three streams %6.3f ms\n", K, gAC, gB, t[0], t[1]);
                    for (int b = 0; b < K; b++) { CK(hipEventDestroy(eA[b])); CK(hipEventDestroy(eB[b])); }
                }
            }
        }
        check("overlapped A2+B2+C2");
    }
    std::vector<uint32_t> h(128);
    CK(hipMemcpy(h.data(), xcc, 512, hipMemcpyDeviceToHost));
    printf("blockIdx%%8 -> XCC_ID histogram (last launch):\n");
    for (int b = 0; b < 8; b++) {
        printf("  b%%8=%d:", b);
        for (int x = 0; x < 8; x++) printf(" %5u", h[b * 16 + x]);
        printf("\n");
    }
    return 0;
}

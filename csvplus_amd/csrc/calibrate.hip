// calibrate.hip — what THIS box's memory system sustains for the two access patterns the Join kernels are made of, so
// that a benchmark can print the ceiling next to the kernel it prices (bench.py: roofline.gather_ceiling_ms):
//   kind 0  streaming copy, 16 bytes per lane, `bytes` read + `bytes` written
//   kind 1  random gather out[i] = table[idx[i]] of 4-byte entries: `n` lookups into a table of `bytes` bytes, 4
//           independent lookups in flight per thread, idx streamed (4 bytes per lookup) and out streamed (4 bytes)
//           — the access pattern of one direct-table step of k_chain_dense without any key decoding
// Not part of the hot path: nothing here is called by IndexOn / Join.
#include "cph_internal.hpp"

namespace cph {

__global__ void k_cal_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

__device__ __forceinline__ uint32_t cal_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_cal_fill_idx(uint32_t* __restrict__ idx, size_t n, uint32_t domain) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        idx[i] = (uint32_t)(((uint64_t)cal_mix((uint32_t)i * 2654435761u + 12345u) * domain) >> 32);
}

template <int R>
__global__ __launch_bounds__(256) void k_cal_gather(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ table,
                                                    uint32_t* __restrict__ out, size_t n) {
    const size_t tile = (size_t)blockIdx.x * 256 * R;
    uint32_t id[R], v[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const size_t i = tile + (size_t)k * 256 + threadIdx.x;
        id[k] = i < n ? idx[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = table[id[k]];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const size_t i = tile + (size_t)k * 256 + threadIdx.x;
        if (i < n) out[i] = v[k];
    }
}

static Status calibrate_run(cph_ctx* ctx, int kind, uint64_t bytes, uint64_t n, int reps, double* ms_out) {
    hipEvent_t a = nullptr, b = nullptr;
    CPH_HIP_TRY(hipEventCreate(&a));
    CPH_HIP_TRY(hipEventCreate(&b));
    struct Ev {
        hipEvent_t a, b;
        ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    } ev{a, b};
    float ms = 0;
    if (kind == 0) {
        DevBuf in, out;
        CPH_TRY(in.alloc(&ctx->pool, bytes));
        CPH_TRY(out.alloc(&ctx->pool, bytes));
        CPH_HIP_TRY(hipMemsetAsync(in.get(), 1, bytes, ctx->stream));
        const size_t nv = bytes / 16;
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_cal_copy, dim3(8192), dim3(256), 0, ctx->stream, in.as<uint4>(), out.as<uint4>(), nv);
        CPH_HIP_TRY(hipEventRecord(a, ctx->stream));
        for (int w = 0; w < reps; w++) hipLaunchKernelGGL(k_cal_copy, dim3(8192), dim3(256), 0, ctx->stream, in.as<uint4>(), out.as<uint4>(), nv);
        CPH_HIP_TRY(hipEventRecord(b, ctx->stream));
        CPH_HIP_TRY(hipEventSynchronize(b));
        CPH_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    } else {
        const uint64_t entries = bytes / 4;
        if (entries == 0 || entries > 0xFFFFFFFFull) return {CPH_ERR_INVALID, "calibrate: table of 1 .. 2^32-1 entries"};
        DevBuf table, idx, out;
        CPH_TRY(table.alloc(&ctx->pool, entries * 4));
        CPH_TRY(idx.alloc(&ctx->pool, n * 4));
        CPH_TRY(out.alloc(&ctx->pool, n * 4));
        CPH_HIP_TRY(hipMemsetAsync(table.get(), 1, entries * 4, ctx->stream));
        hipLaunchKernelGGL(k_cal_fill_idx, dim3(4096), dim3(256), 0, ctx->stream, idx.as<uint32_t>(), (size_t)n, (uint32_t)entries);
        constexpr int R = 4;
        const unsigned grid = (unsigned)((n + 256 * R - 1) / (256 * R));
        for (int w = 0; w < 2; w++)
            hipLaunchKernelGGL(k_cal_gather<R>, dim3(grid), dim3(256), 0, ctx->stream, idx.as<uint32_t>(), table.as<uint32_t>(), out.as<uint32_t>(), (size_t)n);
        CPH_HIP_TRY(hipEventRecord(a, ctx->stream));
        for (int w = 0; w < reps; w++)
            hipLaunchKernelGGL(k_cal_gather<R>, dim3(grid), dim3(256), 0, ctx->stream, idx.as<uint32_t>(), table.as<uint32_t>(), out.as<uint32_t>(), (size_t)n);
        CPH_HIP_TRY(hipEventRecord(b, ctx->stream));
        CPH_HIP_TRY(hipEventSynchronize(b));
        CPH_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    }
    CPH_HIP_TRY(hipGetLastError());
    *ms_out = (double)ms / reps;
    return {};
}

}  // namespace cph

using namespace cph;

extern "C" CPH_API int32_t cph_calibrate(cph_ctx* ctx, int32_t kind, uint64_t bytes, uint64_t n, int32_t reps, double* ms) {
    if (!ctx || !ms || reps < 1 || (kind != 0 && kind != 1) || bytes < 16 || (kind == 1 && n == 0)) return CPH_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail_with(ctx, {CPH_ERR_HIP, "hipSetDevice failed"});
    Status s = calibrate_run(ctx, kind, bytes, n, reps, ms);
    if (!s.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        return fail_with(ctx, s);
    }
    return CPH_OK;
}

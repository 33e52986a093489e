// device_utils.hpp — wave64 / workgroup primitives shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace cph {

constexpr int kWave = 64;  // gfx950 wavefront width (hard-coded on purpose)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// lanes strictly below the caller
__device__ __forceinline__ uint64_t lanemask_lt() {
    return (lane_id() == 0) ? 0ull : (~0ull >> (64 - lane_id()));
}

// Inclusive wave scan (sum) with shuffles.
template <class T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (lane_id() >= d) v += o;
    }
    return v;
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

template <class T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        T o = __shfl_xor(v, d, kWave);
        v = o > v ? o : v;
    }
    return v;
}
template <class T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        T o = __shfl_xor(v, d, kWave);
        v = o < v ? o : v;
    }
    return v;
}

// Workgroup barrier BEHIND LDS ATOMICS.  __syncthreads() makes the compiler wait for a wave's earlier LDS loads and stores, but
// for no-return LDS atomics (atomicAdd / atomicMin / ... on __shared__ memory: ds_add_u32 ...) hipcc 7.0 emits the s_barrier
// without an s_waitcnt lgkmcnt(0): on gfx950 a wave can then signal the barrier while its atomics are still queued, and a wave
// of the same workgroup that reads the counters right behind the barrier sees them short (found in round 4 by the
// differential fuzz: the first-pass histogram of k_encode_split lost ~1 % of its increments in ~1 % of the builds and the
// radix sort scattered into holes).  Every barrier that separates LDS atomics from reads of their targets goes through here.
__device__ __forceinline__ void lds_atomics_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), nothing else waited for
    __syncthreads();
}

// Exclusive workgroup scan of one value per thread.  `smem` needs NWAVES+1
// entries of T.  Returns the exclusive prefix; *total gets the workgroup sum.
// Contains __syncthreads: every thread of the block must call it.
template <class T, int NTHREADS>
__device__ __forceinline__ T block_exclusive_sum(T v, T* smem, T* total) {
    constexpr int NW = NTHREADS / kWave;
    T incl = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) smem[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { T t = smem[w]; smem[w] = run; run += t; }
        smem[NW] = run;
    }
    __syncthreads();
    T res = smem[wave_id()] + incl - v;
    *total = smem[NW];
    __syncthreads();  // smem may be reused by the caller right away
    return res;
}

// ---- reading a string value's bytes with aligned 8-byte loads ------------------------------
// chunk j of a value = its bytes [8j, 8j+8) packed little-endian (byte 8j in bits 0..7); bytes
// past the end of the value are unspecified.  Only the aligned 8-byte words that contain at
// least one byte of the value are touched, so no load can cross into an unmapped page.
// NT: non-temporal loads (the bytes are streamed once; keeps the caches for data that is re-used).
template <bool NT = false>
__device__ __forceinline__ uint64_t load_value_chunk(const uint8_t* data, uint64_t begin, uint64_t len, int j) {
    const uint64_t first = begin + 8ull * (uint64_t)j;                       // byte offset of the chunk
    const uint64_t a = (uint64_t)(uintptr_t)data + first;                    // absolute address
    const uint64_t last = (uint64_t)(uintptr_t)data + begin + (len < 8ull * (j + 1) ? len : 8ull * (j + 1)) - 1;
    // integer -> pointer casts are generic (flat_load); the column bytes live in global memory
    typedef const __attribute__((address_space(1))) uint64_t* global_u64_ptr;
    const global_u64_ptr wp = (global_u64_ptr)(a & ~7ull);
    const int sh = (int)(a & 7ull) * 8;
    uint64_t w0 = NT ? __builtin_nontemporal_load(&wp[0]) : wp[0];
    uint64_t v = w0 >> sh;
    if (sh != 0 && (last & ~7ull) != (a & ~7ull)) v |= (NT ? __builtin_nontemporal_load(&wp[1]) : wp[1]) << (64 - sh);
    return v;
}

// Bytes [8j, 8j+8) of a value, little-endian, with ONE load and WITHOUT any branch.  Bytes past the end of the
// value are unspecified.  Only the aligned 8-byte words that hold at least one byte of the value are touched
// (the rule of load_value_chunk above: no load can run into an unmapped page): a chunk that sits
// inside one aligned word is read as that word and shifted; a chunk that straddles two words is read with one
// UNALIGNED 8-byte load at its first byte (gfx9 global loads take any byte address), which stays inside those
// two words; a chunk that lies entirely past the value reads the word at `base8` (the caller guarantees it is
// readable).  Straight-line code matters here: with a branch per row every key fetch ends in its own s_waitcnt
// and the loads a lane issues for its rows no longer overlap.
//   base8 = a wave-uniform, 8-byte aligned pointer;  x + delta = byte offset of the VALUE from base8 (x is what a
//   lane keeps per row: one 32-bit register when B = u32; the 64-bit address only lives until the load is issued).
template <class B, bool NT = false>
__device__ __forceinline__ uint64_t load_chunk_nobranch(const uint8_t* base8, uint32_t delta, B x, uint32_t len, uint32_t j) {
    typedef const __attribute__((address_space(1))) uint8_t* global_u8_ptr;
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    typedef const __attribute__((address_space(1))) u64_unaligned* global_u64u_ptr;
    const uint32_t off = 8u * j;
    const bool has = len > off;
    const uint32_t left = len - off;
    const uint32_t nb = has ? (left < 8u ? left : 8u) : 1u;
    // a chunk past the value's end reads the word at the value's OWN first byte (never used): always inside the column's allocation
    // (+ its slack).  Round 6: it used to read the column's base pointer — which a chunk column of the streaming Join biases by the
    // chunk's first offset (stream_join.hip: stage_chunk_col), i.e. an address in front of the allocation: a GPU memory fault whenever
    // that page happened to be unmapped (found by the full suite after the allocation pattern of the CSV tests changed).
    const uint64_t a = (uint64_t)x + (has ? delta + off : delta);
    const uint32_t a7 = (uint32_t)a & 7u;
    const bool straddles = a7 + nb > 8u;
    const global_u64u_ptr wp = (global_u64u_ptr)((global_u8_ptr)base8 + (straddles ? a : a & ~7ull));
    // NT: the bytes are streamed once — marked for early eviction so that they do not push re-used data (lookup tables)
    // out of the L2
    const uint64_t w = NT ? __builtin_nontemporal_load(wp) : *wp;
    return w >> (straddles ? 0u : a7 * 8u);
}

__device__ __forceinline__ uint64_t load_offset(const void* offsets, int offset_bits, uint64_t i) {
    return offset_bits == 32 ? (uint64_t) reinterpret_cast<const uint32_t*>(offsets)[i]
                             : reinterpret_cast<const uint64_t*>(offsets)[i];
}

}  // namespace cph

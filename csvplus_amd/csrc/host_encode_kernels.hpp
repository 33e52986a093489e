// host_encode_kernels.hpp — the per-row loops and the worker pool of the host-side key encoder (host_encode.hip), free of any
// HIP or library type so that they compile and are tested on their own (tests/cpp/test_host_encode.cpp, CPU suite).
//
// A key's code is sum over its byte positions p of lut[p][symbol_p], symbol = byte + 1, or 0 when the value has ended
// (keycodec.hip: pre-multiplied LUT; an entry with the top bit set = "not in this position's alphabet").  Two loops:
//   * encode_arith8     one fixed-width 8-byte column over CONTIGUOUS per-position byte ranges (decimal ids): no table —
//                       the bytewise range check of codec_device.hpp (ArithPlan) on one 64-bit load, then 8 multiply-adds;
//   * encode_lut_short  keys of at most 8 byte positions in one column (unpadded decimal ids, short tags): ONE unaligned
//                       8-byte load per value, positions unrolled; the last values of a buffer are read bytewise;
//   * encode_lut        anything else (several columns, longer keys): the plain walk.
// The pool hands out blocks of rows through one monotonically increasing atomic counter (no per-call reset: a late worker
// can never take a block of a job whose description it has not seen), workers spin briefly before they sleep, and the calling
// thread takes blocks too — a chunk of a few million rows is a job of ~1 ms, which a mutex + condition variable per worker
// and call (round 4's first version: 24 fork-joins over 256 threads = 25 ms of pure wake-up) cannot serve.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

namespace cph_host {

constexpr uint32_t kCodeAbsent = 0xFFFFFFFFu;
constexpr int kLutRow = 257;   // symbols per position: 0 = the value ended, 1 + byte

struct HostCol {          // one key column of the chunk (host memory)
    const uint8_t* data;
    const void* offsets;  // nrows + 1 entries (ignored when fixed_width != 0)
    int offset_bits;      // 32 / 64
    uint32_t fixed_width;
    uint64_t data_bytes;  // readable bytes at `data` (0: unknown — the last values are then read bytewise)
};

struct Arith8 {           // encode_arith8's constants
    uint64_t lo, rngc;    // byte p: smallest byte value of position p / 0x7F - (largest - smallest)
    uint32_t mult[8];     // weight of position p
    uint32_t radix[8];    // symbols of position p (largest - smallest + 1): what the vector loop's pair / quad weights are made of
};

inline uint64_t col_offset(const HostCol& c, uint64_t i) {
    return c.offset_bits == 32 ? (uint64_t) static_cast<const uint32_t*>(c.offsets)[i] : static_cast<const uint64_t*>(c.offsets)[i];
}

// The code stores of a block may be NON-TEMPORAL (nt): the codes are read next by the DMA engine that uploads them, never by
// this core — a regular store would first read every output line into the cache (1e8 codes: 400 MB of extra DRAM reads on the
// one NUMA node that holds the pinned buffers, which is what bounds these loops).  Every loop returns whether any row got
// CPH_CODE_ABSENT and fences its non-temporal stores before it returns.
inline void store_code(uint32_t* p, uint32_t v, bool nt) {
#if defined(__x86_64__)
    if (nt) { _mm_stream_si32(reinterpret_cast<int*>(p), (int)v); return; }
#endif
    (void)nt;
    *p = v;
}
inline void store_fence(bool nt) {
#if defined(__x86_64__)
    if (nt) _mm_sfence();
#endif
    (void)nt;
}

inline bool encode_arith8(const Arith8& a, const uint8_t* data, uint64_t r0, uint64_t r1, uint32_t* out, bool nt = false) {
    const uint64_t lo = a.lo, rngc = a.rngc;
    const uint32_t m0 = a.mult[0], m1 = a.mult[1], m2 = a.mult[2], m3 = a.mult[3], m4 = a.mult[4], m5 = a.mult[5], m6 = a.mult[6], m7 = a.mult[7];
    uint64_t any = 0;
    for (uint64_t r = r0; r < r1; r++) {
        uint64_t x;
        memcpy(&x, data + 8 * r, 8);
        // no carries between the bytes of a key that can be in the index; the LOWEST offending byte is always flagged
        const uint64_t z = x - lo, t = z + rngc;
        const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
        const uint32_t code = (zl & 0xFFu) * m0 + ((zl >> 8) & 0xFFu) * m1 + ((zl >> 16) & 0xFFu) * m2 + (zl >> 24) * m3 + (zh & 0xFFu) * m4 +
                              ((zh >> 8) & 0xFFu) * m5 + ((zh >> 16) & 0xFFu) * m6 + (zh >> 24) * m7;
        const uint64_t bad = (x | z | t) & 0x8080808080808080ull;
        any |= bad;
        store_code(out + r, bad ? kCodeAbsent : code, nt);
    }
    store_fence(nt);
    return any != 0;
}

#if defined(__x86_64__)
// The same, four rows per 256-bit vector: bytewise subtract and range check (unsigned max), then the mixed-radix value by
// pmaddubsw (positions (2k, 2k+1) -> 16 bits), pmaddwd (pairs -> the two halves of the key, 32 bits) and one 32-bit multiply-add.
// Needs every odd position's radix <= 127 (a signed byte weight), pair values and pair weights below 2^15: decimal ids and
// anything like them; arith8_vector_ok says whether a codec qualifies.
inline bool arith8_vector_ok(const Arith8& a) {
    if (!__builtin_cpu_supports("avx2")) return false;
    for (int k = 0; k < 4; k++) {
        if (a.radix[2 * k + 1] > 127 || (uint64_t)a.radix[2 * k] * a.radix[2 * k + 1] > 32767) return false;
    }
    return (uint64_t)a.radix[2] * a.radix[3] <= 32767 && (uint64_t)a.radix[6] * a.radix[7] <= 32767;
}
__attribute__((target("avx2"))) inline bool encode_arith8_avx2(const Arith8& a, const uint8_t* data, uint64_t r0, uint64_t r1, uint32_t* out, bool nt = false) {
    alignas(32) uint8_t lo[32], rng[32];
    alignas(32) int8_t w1[32];
    alignas(32) int16_t w2[16];
    for (int i = 0; i < 32; i++) {
        const int p = i & 7;
        lo[i] = (uint8_t)(a.lo >> (8 * p));
        rng[i] = (uint8_t)(a.radix[p] - 1);
        w1[i] = (p & 1) ? 1 : (int8_t)a.radix[p + 1];
    }
    for (int i = 0; i < 16; i++) {
        const int pair = i & 3;   // pair k = positions (2k, 2k+1); quads (0,1) and (2,3)
        w2[i] = (pair & 1) ? 1 : (int16_t)(a.radix[2 * pair + 2] * a.radix[2 * pair + 3]);
    }
    const __m256i LO = _mm256_load_si256((const __m256i*)lo), RNG = _mm256_load_si256((const __m256i*)rng);
    const __m256i W1 = _mm256_load_si256((const __m256i*)w1), W2 = _mm256_load_si256((const __m256i*)w2);
    const uint32_t s_low = a.radix[4] * a.radix[5] * a.radix[6] * a.radix[7];
    const __m256i S = _mm256_set1_epi32((int)s_low), ABSENT = _mm256_set1_epi32(-1);
    uint64_t r = r0;
    __m256i allok = ABSENT;
    bool any = false;
    if (nt) {   // 16-byte streaming stores want 16-byte aligned addresses: the first rows one by one
        uint64_t head = r0;
        while (head < r1 && (reinterpret_cast<uintptr_t>(out + head) & 15u)) head++;
        if (head > r0) any |= encode_arith8(a, data, r0, head, out, true);
        r = head;
    }
    for (; r + 4 <= r1; r += 4) {
        const __m256i x = _mm256_loadu_si256((const __m256i*)(data + 8 * r));
        const __m256i z = _mm256_sub_epi8(x, LO);
        const __m256i ok8 = _mm256_cmpeq_epi8(_mm256_max_epu8(z, RNG), RNG);          // byte in range (a byte below lo wraps above rng)
        const __m256i ok64 = _mm256_cmpeq_epi64(ok8, ABSENT);                         // all 8 bytes of the row
        const __m256i pairs = _mm256_maddubs_epi16(z, W1);
        const __m256i halves = _mm256_madd_epi16(pairs, W2);                          // dword 2i: positions 0..3, dword 2i+1: positions 4..7
        const __m256i hi = _mm256_mullo_epi32(halves, S);                             // (only the even dwords matter)
        const __m256i code = _mm256_add_epi32(hi, _mm256_srli_epi64(halves, 32));     // even dword: p0..3 * S + p4..7
        const __m256i res = _mm256_blendv_epi8(ABSENT, code, ok64);
        // the even dwords of the four 64-bit lanes -> four consecutive u32
        const __m256i packed = _mm256_permutevar8x32_epi32(res, _mm256_setr_epi32(0, 2, 4, 6, 0, 0, 0, 0));
        allok = _mm256_and_si256(allok, ok64);
        if (nt) _mm_stream_si128((__m128i*)(out + r), _mm256_castsi256_si128(packed));
        else _mm_storeu_si128((__m128i*)(out + r), _mm256_castsi256_si128(packed));
    }
    any |= _mm256_movemask_epi8(allok) != -1;
    if (r < r1) any |= encode_arith8(a, data, r, r1, out, nt);
    if (nt) _mm_sfence();
    return any;
}
#else
inline bool arith8_vector_ok(const Arith8&) { return false; }
#endif

// One column, at most 8 positions.  NPOS positions are unrolled; lut = [NPOS][kLutRow].
template <int NPOS>
inline bool encode_lut_short_n(const uint32_t* lut, const HostCol& col, uint64_t r0, uint64_t r1, uint32_t* out, bool nt) {
    const bool fixed = col.fixed_width != 0;
    uint32_t any = 0;
    uint64_t b = fixed ? r0 * (uint64_t)col.fixed_width : (r0 < r1 ? col_offset(col, r0) : 0);
    for (uint64_t r = r0; r < r1; r++) {
        uint64_t e = fixed ? b + col.fixed_width : col_offset(col, r + 1);
        const uint64_t l = e - b;
        uint64_t v = 0;
        if (b + 8 <= col.data_bytes) memcpy(&v, col.data + b, 8);                 // the common case: one load
        else memcpy(&v, col.data + b, l < 8 ? (size_t)l : 8);                     // the buffer's last values
        uint32_t acc = 0, bad = l > (uint64_t)NPOS ? 0x80000000u : 0u;
#pragma GCC unroll 8
        for (int q = 0; q < NPOS; q++) {
            const uint32_t sym = (uint64_t)q < l ? (uint32_t)((v >> (8 * q)) & 0xFFu) + 1u : 0u;
            const uint32_t w = lut[q * kLutRow + sym];
            acc += w;
            bad |= w;
        }
        any |= bad;
        store_code(out + r, (bad >> 31) ? kCodeAbsent : acc, nt);
        b = e;
    }
    store_fence(nt);
    return (any >> 31) != 0;
}

inline bool encode_lut_short(const uint32_t* lut, int npos, const HostCol& col, uint64_t r0, uint64_t r1, uint32_t* out, bool nt = false) {
    switch (npos) {
    case 1: return encode_lut_short_n<1>(lut, col, r0, r1, out, nt);
    case 2: return encode_lut_short_n<2>(lut, col, r0, r1, out, nt);
    case 3: return encode_lut_short_n<3>(lut, col, r0, r1, out, nt);
    case 4: return encode_lut_short_n<4>(lut, col, r0, r1, out, nt);
    case 5: return encode_lut_short_n<5>(lut, col, r0, r1, out, nt);
    case 6: return encode_lut_short_n<6>(lut, col, r0, r1, out, nt);
    case 7: return encode_lut_short_n<7>(lut, col, r0, r1, out, nt);
    default: return encode_lut_short_n<8>(lut, col, r0, r1, out, nt);
    }
}

// Any number of columns and positions: col_start[c] = first position of column c, col_maxlen[c] = its positions.
inline bool encode_lut(const uint32_t* lut, int ncols, const int32_t* col_start, const int32_t* col_maxlen, const HostCol* cols, uint64_t r0,
                       uint64_t r1, uint32_t* out, bool nt = false) {
    uint32_t any = 0;
    for (uint64_t r = r0; r < r1; r++) {
        uint32_t acc = 0, bad = 0;
        for (int c = 0; c < ncols; c++) {
            const HostCol& col = cols[c];
            uint64_t b, l;
            if (col.fixed_width) {
                b = r * (uint64_t)col.fixed_width;
                l = col.fixed_width;
            } else {
                b = col_offset(col, r);
                l = col_offset(col, r + 1) - b;
            }
            const int maxlen = col_maxlen[c];
            if (l > (uint64_t)maxlen) { bad = 0x80000000u; break; }
            const uint32_t* lp = lut + (size_t)col_start[c] * kLutRow;
            const uint8_t* v = col.data + b;
            int q = 0;
            for (; q < (int)l; q++) {
                const uint32_t w = lp[(size_t)q * kLutRow + 1u + v[q]];
                bad |= w;
                acc += w;
            }
            for (; q < maxlen; q++) {   // the value ended: the pad symbol
                const uint32_t w = lp[(size_t)q * kLutRow];
                bad |= w;
                acc += w;
            }
        }
        any |= bad;
        store_code(out + r, (bad >> 31) ? kCodeAbsent : acc, nt);
    }
    store_fence(nt);
    return (any >> 31) != 0;
}

// ---- split codec (keycodec.hip "split codec": prefix dictionary + per-position suffix) ---------------------------------------
// code = rank(prefix) * pmult + sum over suffix positions q of lutw[q][symbol_q]; the prefix is the value through its first `delim`
// byte (the whole value when it holds none), looked up in the codec's perfect hash and verified word for word.  A row the codec
// cannot code (prefix not in the dictionary, a suffix byte or END outside its position's alphabet, lengths beyond the codec's)
// makes the loop return true: the caller then uploads the strings instead (k_encode_split's `miss`).
template <class Key>
struct SplitEnc {
    uint8_t delim;
    uint32_t vmax, smaxlen, pmult, hmask, dmask;
    const uint16_t *disp, *slots;    // perfect hash: slot = ((h >> 16) + disp[h & dmask]) & hmask, slots[slot] = rank + 1 (0: none)
    const Key* dict;                 // .w[4] (bytes little-endian, zero padded), .len
    const uint32_t* lutw;            // [smaxlen][kLutRow]: rank * weight, top bit = not in the alphabet
};

// OFF: the offsets' type; SM: suffix positions as a compile-time bound (0: e.smaxlen at run time) — the position loop unrolls.
template <class Key, class Hash, class OFF, int SM>
inline bool encode_split_t(const SplitEnc<Key>& e, const HostCol& c, uint64_t r0, uint64_t r1, uint32_t* out, bool nt, Hash hash) {
    const uint64_t dv = 0x0101010101010101ull * (uint64_t)e.delim;
    const OFF* off = static_cast<const OFF*>(c.offsets);
    const uint32_t smaxlen = SM ? (uint32_t)SM : e.smaxlen;
    uint32_t any = 0;
    uint64_t end = (uint64_t)off[r0];
    for (uint64_t r = r0; r < r1; r++) {
        const uint64_t b = end;
        end = (uint64_t)off[r + 1];
        const uint64_t l = end - b;
        const uint8_t* p = c.data + b;
        uint64_t w[4], s0, s1;
        uint8_t tmp[48];
        if (l > (uint64_t)e.vmax) { any = 1; store_code(out + r, 0u, nt); continue; }
        if (c.data_bytes == 0 || b + 48 > c.data_bytes) {   // the buffer's last values: through a padded copy
            memset(tmp, 0, sizeof tmp);
            memcpy(tmp, p, (size_t)l);   // (vmax <= 40)
            p = tmp;
        }
        memcpy(w, p, 32);
        // the first delimiter within the first min(l, 32) bytes
        uint32_t at = 64;
        for (int j = 3; j >= 0; j--) {
            const uint64_t x = w[j] ^ dv;
            const uint64_t t = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
            at = t ? 8u * (uint32_t)j + ((uint32_t)__builtin_ctzll(t) >> 3) : at;
        }
        const uint32_t l32 = (uint32_t)l;
        const uint32_t plen = at < l32 ? at + 1u : l32;       // (at >= 32 and l > 32: plen = l > 32 -> a miss below)
        const uint32_t slen = l32 - plen;
        if (plen > 32u || slen > smaxlen) { any = 1; store_code(out + r, 0u, nt); continue; }
        memcpy(&s0, p + plen, 8);                            // (plen + 16 <= 48 readable bytes)
        memcpy(&s1, p + plen + 8, 8);
        for (int j = 0; j < 4; j++) {
            const uint32_t have = plen > 8u * (uint32_t)j ? plen - 8u * (uint32_t)j : 0u;
            w[j] &= have >= 8u ? ~0ull : ((1ull << (8u * have)) - 1ull);
        }
        const uint32_t h = hash(w[0], w[1], w[2], w[3], plen);
        const uint32_t en = e.slots[((h >> 16) + e.disp[h & e.dmask]) & e.hmask];
        const Key& k = e.dict[en ? en - 1 : 0];
        const bool hit = en != 0 && k.len == plen && ((k.w[0] ^ w[0]) | (k.w[1] ^ w[1]) | (k.w[2] ^ w[2]) | (k.w[3] ^ w[3])) == 0;
        uint32_t acc = (hit ? en - 1 : 0u) * e.pmult, bad = hit ? 0u : 0x80000000u;
        const uint32_t* lp = e.lutw;
        for (uint32_t q = 0; q < smaxlen; q++, lp += kLutRow) {
            const uint64_t src = q < 8u ? s0 : s1;
            const uint32_t byte = (uint32_t)(src >> (8u * (q & 7u))) & 0xFFu;
            const uint32_t v = lp[q < slen ? byte + 1u : 0u];
            acc += v & 0x7FFFFFFFu;
            bad |= v;
        }
        any |= bad >> 31;
        store_code(out + r, acc, nt);
    }
    store_fence(nt);
    return any != 0;
}

template <class Key, class Hash>
inline bool encode_split(const SplitEnc<Key>& e, const HostCol& c, uint64_t r0, uint64_t r1, uint32_t* out, bool nt, Hash hash) {
#define CPH_SPLIT_CASE(SM)                                                                                             \
    case SM:                                                                                                           \
        return c.offset_bits == 32 ? encode_split_t<Key, Hash, uint32_t, SM>(e, c, r0, r1, out, nt, hash)              \
                                   : encode_split_t<Key, Hash, uint64_t, SM>(e, c, r0, r1, out, nt, hash);
    switch (e.smaxlen <= 8u ? e.smaxlen : 0u) {
        CPH_SPLIT_CASE(1) CPH_SPLIT_CASE(2) CPH_SPLIT_CASE(3) CPH_SPLIT_CASE(4) CPH_SPLIT_CASE(5) CPH_SPLIT_CASE(6) CPH_SPLIT_CASE(7) CPH_SPLIT_CASE(8)
        default: break;
    }
#undef CPH_SPLIT_CASE
    return c.offset_bits == 32 ? encode_split_t<Key, Hash, uint32_t, 0>(e, c, r0, r1, out, nt, hash)
                               : encode_split_t<Key, Hash, uint64_t, 0>(e, c, r0, r1, out, nt, hash);
}

// ---- worker pool ----------------------------------------------------------------------------------------------------------
// run(nrows, fn): fn(r0, r1) is called for disjoint row blocks that cover [0, nrows), on the workers and on the calling thread;
// returns when every block is done.  One job at a time (the caller serialises).
class BlockPool {
public:
    static constexpr uint64_t kBlockRows = 1u << 16;

    explicit BlockPool(int nworkers) {
        for (int i = 0; i < nworkers; i++) workers_.emplace_back([this] { worker(); });
    }
    ~BlockPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    BlockPool(const BlockPool&) = delete;
    BlockPool& operator=(const BlockPool&) = delete;
    int workers() const { return (int)workers_.size(); }
#if defined(__linux__)
    // the workers may only run on these CPUs from now on (the caller's thread is not touched)
    void set_affinity(const cpu_set_t& set) {
        for (auto& t : workers_) (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
    }
#endif

    template <class F>
    void run(uint64_t nrows, F&& fn, uint64_t block_rows = kBlockRows) {
        if (nrows == 0) return;
        if (block_rows == 0) block_rows = kBlockRows;
        const uint64_t nblocks = (nrows + block_rows - 1) / block_rows;
        if (nblocks == 1 || workers_.empty()) {
            fn((uint64_t)0, nrows);
            return;
        }
        Fn<F> job{&fn};
        job_call_ = &Fn<F>::call;
        job_ctx_ = &job;
        job_rows_ = nrows;
        job_block_ = block_rows;
        base_ = limit_.load(std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        limit_.store(base_ + nblocks, std::memory_order_seq_cst);   // publishes the job: blocks [base_, base_ + nblocks)
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {   // (seq_cst on both sides: a worker about to sleep sees the job or is seen)
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
        take_blocks();
        while (done_.load(std::memory_order_acquire) != nblocks) cpu_relax();
    }

private:
    template <class F>
    struct Fn {
        F* f;
        static void call(void* self, uint64_t r0, uint64_t r1) { (*static_cast<Fn*>(self)->f)(r0, r1); }
    };
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // Takes blocks until none is left.  A block id below limit_ belongs to the job limit_'s store published (acquire).
    bool take_blocks() {
        bool any = false;
        for (;;) {
            uint64_t b = next_.load(std::memory_order_relaxed);
            if (b >= limit_.load(std::memory_order_acquire)) return any;
            if (!next_.compare_exchange_weak(b, b + 1, std::memory_order_acq_rel)) continue;
            const uint64_t i = b - base_;
            const uint64_t r0 = i * job_block_, r1 = r0 + job_block_ < job_rows_ ? r0 + job_block_ : job_rows_;
            job_call_(job_ctx_, r0, r1);
            done_.fetch_add(1, std::memory_order_acq_rel);
            any = true;
        }
    }
    void worker() {
        for (;;) {
            if (take_blocks()) continue;
            // nothing to do: spin for a while (the next chunk usually follows within microseconds), then sleep
            bool woke = false;
            for (int spin = 0; spin < 4000 && !woke; spin++) {
                cpu_relax();
                woke = next_.load(std::memory_order_relaxed) < limit_.load(std::memory_order_acquire) || quit_.load(std::memory_order_acquire);
            }
            if (quit_.load(std::memory_order_acquire)) return;
            if (woke) continue;
            std::unique_lock<std::mutex> lk(mu_);
            sleepers_.fetch_add(1, std::memory_order_seq_cst);
            cv_.wait(lk, [&] { return quit_.load(std::memory_order_acquire) || next_.load(std::memory_order_relaxed) < limit_.load(std::memory_order_seq_cst); });
            sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            if (quit_.load(std::memory_order_acquire)) return;
        }
    }

    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> quit_{false};
    std::atomic<int> sleepers_{0};
    alignas(64) std::atomic<uint64_t> next_{0};
    alignas(64) std::atomic<uint64_t> limit_{0};
    alignas(64) std::atomic<uint64_t> done_{0};
    // the current job (written by run() before limit_ is raised; read by whoever holds one of its blocks)
    void (*job_call_)(void*, uint64_t, uint64_t) = nullptr;
    void* job_ctx_ = nullptr;
    uint64_t job_rows_ = 0, job_block_ = kBlockRows, base_ = 0;
};

}  // namespace cph_host

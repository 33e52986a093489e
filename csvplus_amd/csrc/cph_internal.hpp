// cph_internal.hpp — internal structures of libcsvplus_hip (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/csvplus_hip.h"

namespace cph {

// ---- limits / launch geometry ------------------------------------------------
constexpr int kMaxKeyBytes = 128;                // byte positions one codec (one key WINDOW) covers; longer keys take
                                                 // several windows (cph_index::windows)
constexpr int kMaxKeyCols  = CPH_MAX_KEY_COLS;
constexpr int kLutStride   = 257;                // symbols per position: pad + 256 byte values
constexpr uint16_t kLutInvalid = 0xFFFF;
constexpr int kMaxWords    = 20;                 // code words per window (each < 2^63 states; 128 positions of a
                                                 // full 257-symbol alphabet need 19)

// ---- error plumbing ------------------------------------------------------------
struct Status {
    int32_t code = CPH_OK;
    std::string msg;
    bool ok() const { return code == CPH_OK; }
};

#define CPH_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char buf_[512];                                                                        \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                     __FILE__, __LINE__);                                                          \
            return ::cph::Status{CPH_ERR_HIP, buf_};                                               \
        }                                                                                          \
    } while (0)

#define CPH_TRY(expr)                        \
    do {                                     \
        ::cph::Status s_ = (expr);           \
        if (!s_.ok()) return s_;             \
    } while (0)

// ---- device memory pool -----------------------------------------------------------
// All work of a ctx runs on one stream, so a block freed (in host program order)
// after the kernels using it were enqueued can be handed to the next user: reuse is
// stream-ordered.  Blocks are cached until cph_ctx_destroy / trim().
// The bookkeeping is locked: a Join running on another ctx (another thread) may build a lookup structure of an
// index, and that allocates from the pool of the INDEX's ctx (probe.hip: accel_ctx).
class DevicePool {
public:
    Status alloc(size_t bytes, void** out);
    void   release(void* p);   // returns the block to the cache
    void   trim();             // hipFree everything cached
    ~DevicePool();
    size_t bytes_live = 0, bytes_cached = 0, n_hipmalloc = 0;
    // Optional slab (cph_ctx_set_option "pool_reserve_mb"): ONE hipMalloc up front, blocks carved out of it first-fit
    // and coalesced on release — a one-shot caller then pays no hipMalloc (tens of ms per GB on a cold device) inside
    // its first call.  Requests the slab cannot serve fall through to the per-size cache below.
    Status reserve(size_t bytes);
    // guard mode (debugging aid): canary bytes behind every block, verified at release
    bool guard = false;
    uint64_t guard_violations = 0;
    std::string first_violation;
    void check_live();
    // While a batch of builds runs on TWO streams (capi.hip: cph_index_build_many), reuse is no longer stream-ordered: a
    // block released by one build could be handed to the other while kernels still use it.  Between begin_defer and
    // end_defer released blocks are parked instead; end_defer (called once both streams are idle) really releases them.
    void begin_defer();
    void end_defer();
    // Both streams of the batch are idle (the caller has just synchronised them): the parked blocks go back to the cache now, so
    // that the batch's peak footprint stays one phase's worth instead of the sum over all jobs; parking goes on afterwards.
    void flush_deferred();

private:
    struct Block { void* p; size_t cap; size_t user; bool guarded = false; bool in_slab = false; };
    void check_block(const Block& b);
    std::recursive_mutex mu_;
    uint8_t* slab_ = nullptr;
    size_t slab_bytes_ = 0;
    std::vector<std::pair<size_t, size_t>> slab_free_;   // (offset, length), sorted by offset, coalesced
    std::vector<Block> free_;
    std::vector<Block> live_;
    int defer_depth_ = 0;
    std::vector<void*> deferred_;
};

// RAII handle on a pool block.
class DevBuf {
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept { *this = std::move(o); }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { reset(); pool_ = o.pool_; p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.pool_ = nullptr; o.bytes_ = 0; }
        return *this;
    }
    ~DevBuf() { reset(); }
    Status alloc(DevicePool* pool, size_t bytes) {
        reset();
        pool_ = pool;
        bytes_ = bytes;
        return pool->alloc(bytes ? bytes : 1, &p_);
    }
    void reset() {
        if (p_ && pool_) pool_->release(p_);
        p_ = nullptr;
    }
    template <class T> T* as() const { return static_cast<T*>(p_); }
    void* get() const { return p_; }
    size_t bytes() const { return bytes_; }
    explicit operator bool() const { return p_ != nullptr; }

private:
    DevicePool* pool_ = nullptr;
    void* p_ = nullptr;
    size_t bytes_ = 0;
};

}  // namespace cph

namespace cph_host {
struct HostCol;
class BlockPool;
}  // namespace cph_host

// ---- key codec description (host + device copies) -------------------------------------
namespace cph {

// A whole value of at most 32 bytes as a dictionary key: its bytes little-endian in four words, zero padded, + its length.
constexpr int kWideBytes = 32;
constexpr int kWideDictMax = 1024;          // distinct prefixes a split codec takes (40 KiB of LDS)
struct WideKey {
    uint64_t w[4];
    uint32_t len;
    uint32_t pad_;
};
#if defined(__HIPCC__)
#define CPH_HD2 __host__ __device__
#else
#define CPH_HD2
#endif
// 64-bit hash of a WideKey (host and device agree): the low half picks the slot of the codec block's lookup table, the
// whole value is the tag under which the statistics pass collects the distinct prefixes.  Never 0 (0 = empty slot).
CPH_HD2 inline uint32_t wide_rotl(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
// (two functions: the kernel that only looks prefixes up pays for the low half alone)
CPH_HD2 inline void wide_fold(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len, uint32_t* x, uint32_t* y) {
    *x = (uint32_t)w0 ^ wide_rotl((uint32_t)w1, 9) ^ wide_rotl((uint32_t)w2, 18) ^ wide_rotl((uint32_t)w3, 27) ^ (len * 0x01000193u);
    *y = (uint32_t)(w0 >> 32) ^ wide_rotl((uint32_t)(w1 >> 32), 7) ^ wide_rotl((uint32_t)(w2 >> 32), 14) ^ wide_rotl((uint32_t)(w3 >> 32), 21);
}
CPH_HD2 inline uint32_t wide_hash_lo(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) {
    uint32_t x, y;
    wide_fold(w0, w1, w2, w3, len, &x, &y);
    uint32_t h1 = (x * 0x9E3779B1u) ^ (y * 0x85EBCA6Bu);
    h1 ^= h1 >> 15;
    return h1;
}
CPH_HD2 inline uint64_t wide_hash(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) {
    uint32_t x, y;
    wide_fold(w0, w1, w2, w3, len, &x, &y);
    const uint32_t h1 = wide_hash_lo(w0, w1, w2, w3, len);
    uint32_t h2 = (x * 0xCC9E2D51u + wide_rotl(y, 13)) * 0x1B873593u;
    h2 ^= h2 >> 16;
    h2 *= 0xC2B2AE35u;
    h2 ^= h2 >> 13;
    const uint64_t h = (uint64_t)h1 | ((uint64_t)h2 << 32);
    return h ? h : 1ull;
}

// How the key columns map to the mixed-radix code (see keycodec.hip).
struct CodecHost {
    int32_t ncols = 0;
    int32_t npos = 0;                       // total byte positions (sum of maxlen)
    int32_t nwords = 1;                     // code words per key
    bool    key32 = false;                  // single word that fits 32 bits
    int32_t col_start[kMaxKeyCols + 1] = {0};   // first position of each column
    int32_t col_maxlen[kMaxKeyCols] = {0};
    int32_t col_minlen[kMaxKeyCols] = {0};
    int32_t word_bits[kMaxWords] = {0};     // significant bits per word
    uint64_t word_states[kMaxWords] = {0};  // number of states per word (product of radices)
    // per position
    std::vector<uint16_t> radix;            // [npos]
    std::vector<uint64_t> mult;             // [npos] weight inside its word
    std::vector<int32_t>  word_of;          // [npos]
    std::vector<uint16_t> lut;              // [npos][257] rank or kLutInvalid
    // Dictionary-coded groups (only used when the per-position code would need several words): a group is up
    // to kGroupSpan consecutive byte positions of one column whose bytes JOINTLY take few distinct values in the
    // build table.  The head position carries radix = #distinct; its rank is the index of the window's raw key
    // (group_raw) in `dict`, which lists the raw keys in the order of their position tuples (group_order_key);
    // the positions behind the head are absorbed (radix 1, every symbol ok).
    std::vector<uint8_t>  unit;             // [npos] kUnitPos / kUnitHead / kUnitAbsorbed (empty: no groups)
    std::vector<int32_t>  dict_off;         // [npos] head: first entry of its dictionary in `dict`
    std::vector<int32_t>  dict_len;         // [npos] head: number of entries
    std::vector<uint64_t> dict;             // raw keys of all heads, each head's in rank order
    bool has_groups() const { return !unit.empty(); }
    // Delimiter split (round 4, keycodec.hip "split codec"): key column `split_col` of the TABLE is coded as two virtual
    // columns — the prefix up to and including the first byte `split_byte` (the whole value when it holds none) and the
    // suffix behind it.  No prefix is a proper prefix of another one that ends in the delimiter, so comparing
    // (prefix, suffix) tuples with strings.Compare equals comparing the values (csvplus.go:794-807 order kept).
    // ncols / col_start / col_maxlen / col_minlen then describe the VIRTUAL columns (one more than the index has key
    // columns: codec_virtual_cols maps), the prefix column is one kUnitWide head (radix = number of distinct prefixes,
    // rank = index into wdict, which lists them in strings.Compare order) followed by absorbed positions, the suffix
    // column is coded per position like any column.  Fields whose digits FLOAT behind a variable-length head
    // ("Smith/Amelia#12345") cost their information, not their byte positions: BASELINE config 3 codes in 25 bits.
    int32_t split_col = -1;                 // key column of the table that is split (-1: none)
    uint8_t split_byte = 0;
    bool spec_checked = false;              // split codec built from a SAMPLE (codec_try_split speculate): the build's encode kernel checks suffix bytes too
    int32_t split_maxlen = 0;               // longest value of the split column in the build table (which kernel instantiation encodes it)
    std::vector<WideKey> wdict;             // distinct prefixes in rank order
    // PERFECT hash of wdict (codec_wide_perfect_hash; derived, not serialised): a lookup is h = wide_hash_lo(key),
    // slot = ((h >> 16) + wide_disp[h & (nbuckets - 1)]) & (nslots - 1), wide_slots[slot] = rank + 1 (0: no such key) —
    // no probe sequence, so a wave never loops over its unluckiest lane
    std::vector<uint16_t> wide_disp, wide_slots;
    bool has_split() const { return split_col >= 0; }
    int32_t virtual_cols(int32_t real_cols) const { return real_cols + (has_split() && split_col < real_cols ? 1 : 0); }
};
constexpr int kGroupSpan = 7;               // positions per group: 7 bytes + a length fit one 64-bit raw key
constexpr int kGroupDictMax = 4096;         // dictionary entries per index (32 KiB of LDS)
enum : uint8_t { kUnitPos = 0, kUnitHead = 1, kUnitAbsorbed = 2, kUnitWide = 3 };   // kUnitWide: head of a whole-column dictionary (split codec)

// Device-side codec block, laid out for one cooperative copy into LDS:
//   header (CodecDevHeader) | mult[npos] u64 | word_of[npos] u8 (padded) | lut[npos*257] u16
struct CodecDevHeader {
    int32_t ncols;
    int32_t npos;
    int32_t nwords;
    int32_t key32;
    int32_t col_start[kMaxKeyCols + 1];
    int32_t col_maxlen[kMaxKeyCols];
    int32_t mult_off;      // byte offsets from the start of the block
    int32_t wordof_off;
    int32_t lut_off;       // rank LUT u16[npos][257]; 0 when the pre-multiplied LUT replaces it
    int32_t total_bytes;   // multiple of 16
    // pre-multiplied LUT (single-word codes, when it fits): lutw[p][sym] = rank * mult[p], so the
    // code is a plain sum of one LDS load per byte position; the top bit marks "symbol not in the
    // alphabet".  lutw_bits = 0 (absent), 32 or 64.
    int32_t lutw_off;
    int32_t lutw_bits;
    // dictionary-coded groups (all 0 when there are none): unit u8[npos], dict_off/dict_len i32[npos], dict u64[]
    int32_t ngroups;
    int32_t unit_off;
    int32_t dictoff_off;
    int32_t dictlen_off;
    int32_t dict_off;
    // open-addressing lookup of a raw key: slot = group_slot(raw, hash_bits[p]); hash u16[] holds rank + 1
    // (0 = empty) at hash_off[p] + slot, verified against dict[]; built by codec_upload
    int32_t hashoff_off;   // i32[npos]
    int32_t hashbits_off;  // i32[npos]
    int32_t hash_off;      // u16[]
    // split codec (all 0 / -1 when there is none): the whole-column dictionary of the prefix column
    int32_t wide_pos;      // position of the kUnitWide head, -1: none
    int32_t wide_n;        // entries
    int32_t wide_off;      // WideKey[wide_n], rank order
    int32_t wide_hash_off; // u16[1 << wide_hash_bits]: rank + 1 (0 = empty); perfect hash (CodecHost::wide_disp)
    int32_t wide_hash_bits;
    int32_t split_vcol;    // virtual column that is the prefix (its successor is the suffix), -1: none
    int32_t split_byte;
    int32_t wide_disp_off; // u16[1 << wide_disp_bits]: displacement of each bucket
    int32_t wide_disp_bits;
    int32_t pad_[3];
};
#if defined(__HIPCC__)
#define CPH_HD __host__ __device__
#else
#define CPH_HD
#endif
// Slot of a raw key in a head's hash table of 2^bits slots (host and device must agree).
CPH_HD inline uint32_t group_slot(uint64_t raw, int bits) { return (uint32_t)((raw * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }
// Raw key of a group window: `window` holds the value's bytes from the group's first position on (byte i of the
// window = value byte q0 + i, anything past the value's end), nvalid = how many of the group's positions the value
// still covers (0..span <= 7).  Injective in (nvalid, the valid bytes): cheap to form, used for equality only.
CPH_HD inline uint64_t group_raw(uint64_t window, uint64_t nvalid) {
    return (window & ((1ull << (8 * nvalid)) - 1ull)) | (nvalid << 56);
}
// Order key of a raw key: 9 bits per position (0 = the value ended, 1 + byte), first position on top — compares
// like the tuple of the group's position symbols, i.e. like strings.Compare on that stretch of the key.
CPH_HD inline uint64_t group_order_key(uint64_t raw, int span) {
    const int nvalid = (int)(raw >> 56);
    uint64_t k = 0;
    for (int i = 0; i < span && i < nvalid; i++) k |= (((raw >> (8 * i)) & 0xFFull) + 1ull) << (9 * (kGroupSpan - 1 - i));
    return k;
}

}  // namespace cph

// ---- optional per-kernel timing (HIP events on the ctx stream) -------------------------------
namespace cph {
struct ProfPending { int name_idx; hipEvent_t start, stop; double bytes; };
struct ProfStat { std::string name; uint64_t launches = 0; double total_ms = 0, bytes = 0; };
}  // namespace cph

// ---- the opaque C types ------------------------------------------------------------------
struct cph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    cph::DevicePool pool;
    // small pinned scratch for read-backs
    void* pinned_scratch = nullptr;
    size_t pinned_scratch_bytes = 0;
    // pinned ring for small host->device uploads (codec blocks): bump-allocated, the stream is only
    // synchronised when the ring wraps, so an upload never stalls the pipeline
    void* upload_ring = nullptr;
    size_t upload_cap = 0, upload_pos = 0;
    std::vector<void*> pinned_user;
    // small blocks handed back through cph_pinned_free are kept for the next cph_pinned_alloc (a host side that stages a batch of
    // 8192 rows per call page-locks four blocks per batch otherwise: 4 x 90 us against 60-80 us for the Join itself): blocks of at
    // most 4 MB, at most 32 of them; {block, capacity}
    std::vector<std::pair<void*, size_t>> pinned_user_free;
    std::vector<std::pair<void*, size_t>> pinned_user_cap;   // capacity of every live cph_pinned_alloc block that came from / may go to the list
    // pinned blocks of released host-side results (cph_index_perm copies), reused by the next one of a fitting size:
    // page-locking 400 MB costs tens of milliseconds, a caller that builds index after index should pay it once
    std::vector<std::pair<void*, size_t>> pinned_cache;
    // pinned, device-visible words for what kernels REPORT to the host: a kernel stores into them directly (plain stores over
    // PCIe: rare, a flag or a total), the host reads them after its next synchronisation — no memset to arm one (the host
    // writes it), no device-to-host copy to read it.  Taken round robin (host_word): a word is in use for one call only.
    uint32_t* host_words = nullptr;
    uint32_t host_words_pos = 0;
    // Device accumulators that CLEAN UP AFTER THEMSELVES (zeroed once when allocated; the last workgroup of the kernel that
    // fills one exports the result to a report word and zeroes it again): no memset in front of the kernel, no copy behind it.
    // One set per stream slot (the two streams of a build batch run concurrently).
    struct SelfClean {
        cph::DevBuf sum;           // k_sum_counts_report: {u64 total, u32 ticket, u32 pad}
        cph::DevBuf sample;        // k_split_count of a sampled build: SplitSample + ticket
        cph::DevBuf win;           // window_sort.hip: bucket cursors
        uint64_t win_words = 0;
    } self_clean[2];
    cph::DevBuf safe_words;        // 64 readable device bytes: the data base of columns that come without a buffer
    // per-ctx launch state (a process may hold one ctx per device: nothing of this may be static)
    int cus = 0;                   // compute units of `device` (0: not queried yet)
    int chain_debug = 0;           // attribution switches of the chained-join kernel (cph_ctx_set_option)
    int sort_threads = 0, sort_rbits = 0;   // radix-sort tuning overrides (0: automatic)
    int sort_digit_stream = 1;     // scatter writes the next pass's digits as a byte stream for its histogram (radix_sort.hip)
    int sort_xcd_tiles = 1;        // scatter: contiguous tile ranges per XCD (radix_sort.hip)
    int small_build_rows = 8192;   // tables of at most this many rows (<= 16384) are indexed by ONE launch of one workgroup (small_build.hip); 0: never
    int stream_role_streams = 0;   // cph_stream_join (fused mode): 1 = one stream for all uploads, one for all downloads; 0 (default, faster at 2 and 4 slots): everything of a slot on its own stream
    int stream_zero_copy_out = 0;  // cph_stream_join (fused mode): the kernel stores the row ids straight into the slot's pinned block
    int chain_nt_streams = 0;      // chained join: non-temporal loads / stores for the stream's bytes and the results (0 never, 1 always, 2 positions mode)
    int chain_arith = 1;           // chained join: fixed-width key columns over contiguous alphabets are encoded arithmetically (codec_device.hpp: ArithPlan) and read with one aligned load (A/B switch)
    int chain_identity = 1;        // positions mode: an index whose code space is exactly as large as the index needs no lookup (A/B switch)
    int chain_rows4 = 1;           // chained join: register-heavy kernel variants walk 4 rows per lane and phase instead of 8 (0: never, 2: every non-lean chain; A/B switch)
    int chain_rank_lds = 1;        // positions mode: rank tables of small indexes are copied into LDS by every workgroup (A/B switch)
    int probe_hash_rows = 2;       // rows per phase of the generic hash probe (2 / 4): 4 rows need 164 VGPRs (3 waves per SIMD) and measured 20 % slower
    int join_hash = 1;             // 0: indexes of this ctx never get a hash table (A/B switch: sorted search instead)
    int codec_debug = 0;           // prints the window choice of codec_try_groups to stderr
    uint64_t n_split_respec = 0;   // builds whose sampled split codec missed a row and that started over with the exact statistics (cph_ctx_get_stat)
    int hash_load_pct = 50;        // load factor of the Join hash tables, per cent of a sector's slots (probe.hip: index_ensure_hash)
    int direct_fused_encode = 1;   // the direct sort of fixed-width 8-byte ids codes the keys inside its first partition level (no encode kernel; A/B switch)
    int direct_ranktab = 1;        // the window sort writes the index's rank table while it streams its windows out (no k_build_ranktab at the first Join; A/B switch)
    int chain_prejoin = 1;         // chain steps keyed by an earlier build table are answered from pre-joined tables (chain.hip: run_prejoined; 0: the DEP kernel)
    int split_speculative = 1;     // the split codec of a large single-column table is taken from its sample, checked by the encode kernel (0: exact pass)
    int codec_split = 1;           // the delimiter split of keycodec.hip is tried (A/B switch; 0: never)
    int scan_lookback = 1;         // exclusive_scan_u32 as ONE launch (decoupled look-back, radix_sort.hip) instead of three (A/B switch)
    struct ScanState {             // its state words (+ ticket counter), zeroed once; epochs / relative tickets make a scan memset-free
        cph::DevBuf words;
        uint64_t tiles = 0, epoch = 0;
        uint32_t tickets = 0;
    } scan[2];                     // one per stream slot (scans of the two streams of a build batch run concurrently)
    int csv_fast = 1;              // cph_csv_parse: byte-parallel passes over text tiles for texts without quotes (csv_ingest.hip: k_csv_fast); 0: the record-parallel kernels
    int csv_onepass = 1;           // cph_csv_write[_rows]: one pass over the joined rows (materialize.hip: k_csv_onepass; slot tables + decoupled look-back);
                                   // 0: the two-pass writer; N > 1: taken whatever the row count, with at most N workgroups (tests)
    int csv_onepass_debug = 0;     // measurement only (wrong text): 1 no look-back, 2 no record bytes, 4 one record per thread
    int sample_lean = 1;           // the sample of a fixed-width key column of <= 8 bytes is taken by k_sample_fixed8 (A/B switch; 0: k_split_count)
    int hash_partitioned = 1;      // the hash table of a duplicate-free index of >= 2^21 keys is built slice by slice in LDS (probe.hip); 0: CAS into the whole
                                   // table; 2: slice by slice whatever the size (tests)
    int counted_sort = 1;          // IndexOn over 32-bit codes with duplicates: MSD sort through counted LDS windows (counted_sort.hip); 0: the classic passes
    int direct_sort = 1;           // a build that expects distinct keys (UniqueIndexOn) over a dense 32-bit code space (rows <= states <= 2 rows)
                                   // sorts by ONE scatter, slot[code] = row (radix_sort.hip: direct_sort_distinct); a duplicate is noticed on
                                   // the device and the build starts over the general way (A/B switch)
    int stats_sample = 1;          // IndexOn over ONE fixed-width key column of >= 2^20 rows takes its alphabets from a sample; the encode
                                   // kernel checks every row against them and the build starts over with exact statistics on a miss (A/B switch)
    int host_build = 1;            // cph_index_build over ONE key column of <= 8 byte positions in HOST memory: the codes are formed by host
                                   // threads and only they are uploaded (host_encode.hip: build_from_host_codes); 0: always upload the strings
    int host_threads = 0;          // threads of the ctx's host worker pool (0: half the hardware threads, at most 32)
    void* host_pool = nullptr;     // cph::HostPool, created on first use (host_encode.hip)
    int host_split = 1;            // ... and a variable-length column the delimiter split codes (config 3's keys): the split codec's host twin — 1: when the
                                   // host's threads (within the cgroup's CPU quota) beat the upload of the strings by the estimate, 2: always, 0: never
    int host_split_threads = 0;    // threads of that (compute-bound) loop's own pool (0: 3/8 of the hardware threads, at most 96, at most the CPU quota)
    void* host_pool_wide = nullptr;
    int host_numa = 0;             // 1: that pool's workers are bound to the NUMA node that holds the column (opt-in: its effect could not be measured)
    int build_side_stream = 1;     // cph_index_build_many: every second general build of a batch runs on a second stream (A/B switch)
    hipStream_t side_stream = nullptr;   // created on first use
    hipEvent_t side_fork = nullptr;      // recorded on `stream` when a two-stream batch starts; side_stream waits for it, so that the
                                         // side jobs are ordered behind everything the caller had enqueued on the ctx's stream
    int stream_slot = 0;           // 0: `stream` is the ctx's own; 1: it is side_stream for the moment (cph::SideStream)
    hipStream_t swapped_main = nullptr;  // ... and this is the ctx's own meanwhile
    hipStream_t other_stream() const { return stream_slot ? swapped_main : side_stream; }   // may be null
    int plan_threads = 0, gstats_threads = 0;   // tuning: workgroup sizes of k_encode_build_plan / k_group_stats (0: default)
    int speculative_groups = 1;    // dictionaries of large inputs from a sample, completed by the encode kernel (keycodec.hip):
                                   // 0 never, 1 when the sample holds no value seen only once, 2 always
    struct KernelCfg { const void* fn; size_t lds; int blocks_per_cu; };
    std::vector<KernelCfg> kernel_cfg;   // kernels whose dynamic-LDS attribute / occupancy were set up on this device
    // profiling
    bool profiling = false;
    std::string prof_only;         // non-empty: only launches of this kernel are timed (cph_ctx_profile_only)
    std::vector<cph::ProfPending> prof_pending;
    std::vector<cph::ProfStat> prof_stats;
    std::vector<hipEvent_t> prof_free_events;
};

namespace cph {
constexpr uint32_t kHostWords = 65536;
// n consecutive report words (8-byte aligned), zeroed by the host; nullptr when the block could not be allocated.  The ring
// wraps without looking: a word is in use for ONE API call only, so a call that takes many of them (a batch of builds: up to
// 64 jobs x one SplitSample each) first makes sure they fit in front of the wrap point (host_words_reserve) — no word handed out
// inside a call is ever handed out again before the call returns.
uint32_t* host_word(cph_ctx* ctx, uint32_t n = 1);
// at an API call's entry (nothing outstanding): the next `n` words will come without a wrap; false if n can never fit
bool host_words_reserve(cph_ctx* ctx, uint32_t n);
// a zeroed device block of at least `bytes` for a self-cleaning accumulator (allocated / grown on demand: one memset and
// one synchronisation, once)
Status self_clean_block(cph_ctx* ctx, cph::DevBuf* b, size_t bytes);
// Everything a ctx enqueues goes to ctx->stream; for the duration of this guard that is the side stream.
struct SideStream {
    cph_ctx* ctx;
    hipStream_t saved;
    SideStream(cph_ctx* c, bool on) : ctx(on ? c : nullptr), saved(c->stream) {
        if (ctx) {
            ctx->swapped_main = saved;
            ctx->stream = ctx->side_stream;
            ctx->stream_slot = 1;
        }
    }
    ~SideStream() {
        if (ctx) {
            ctx->stream = saved;
            ctx->stream_slot = 0;
            ctx->swapped_main = nullptr;
        }
    }
    SideStream(const SideStream&) = delete;
    SideStream& operator=(const SideStream&) = delete;
};
}  // namespace cph

// One window of at most kMaxKeyBytes key byte positions (positions run column-major over the key columns, so a
// window is a list of column SEGMENTS).  Keys that fit one window — every key the tuned paths ever see — have
// cph_index::windows empty and live in cph_index::codec alone.
struct cph_key_window {
    cph::CodecHost codec;          // built over the window's segments as if they were the key columns
    cph::DevBuf codec_dev;
    int32_t nseg = 0;
    int32_t seg_col[CPH_MAX_KEY_COLS] = {0};
    uint32_t seg_skip[CPH_MAX_KEY_COLS] = {0}, seg_take[CPH_MAX_KEY_COLS] = {0};   // take 0xFFFFFFFF: the column's last segment
    int32_t word_base = 0;         // its first word in sorted_codes (windows and words in significance order)
};

struct cph_index {
    cph_ctx* ctx = nullptr;
    uint64_t nrows = 0;
    uint64_t table_rows = 0;       // rows of the table the index was built over (perm values are < table_rows)
    int32_t nkeycols = 0;
    cph::CodecHost codec;          // single-window keys: THE codec; several windows: a copy of windows[0].codec
    cph::DevBuf codec_dev;         // CodecDevHeader block (single-window keys)
    std::vector<cph_key_window> windows;   // non-empty only when the key columns need more than kMaxKeyBytes positions
    int32_t total_words() const {
        if (windows.empty()) return codec.nwords;
        int32_t t = 0;
        for (const auto& w : windows) t += w.codec.nwords;
        return t;
    }
    cph::DevBuf sorted_codes;      // key32: u32[n]; else u64[total_words()][n] word-major
    cph::DevBuf perm;              // u32[n]
    // direct-address tables over the code space, built by the first Join that can use them (probe.hip)
    cph::DevBuf table;             // {lo,row} / {lo,end} u32x2 [table_entries]: generic probe
    cph::DevBuf rowtab;            // duplicate-free index: u32[table_entries], code -> build row (0xFFFFFFFF: absent);
                                   // 4-byte entries for the chained-join kernel (half the random-access footprint)
    cph::DevBuf ranktab;           // duplicate-free index: {u32 present bits, u32 keys before} per 32 codes — code -> SORTED
                                   // POSITION (rank) of the key, 1/16 of rowtab's footprint: what a Join that reports index
                                   // positions (cph_join_chain_ex CPH_CHAIN_POSITIONS) looks up
    uint64_t table_entries = 0;    // != 0: the code space is dense enough for a table (decided at build time)
    // hash table over the codes for every other index (hash_device.hpp), built by the first full-key Join
    cph::DevBuf hash;              // hash_sectors x 64 bytes
    uint32_t hash_sectors = 0;
    uint32_t hash_slice_mask = 0;  // != 0: the table was built slice by slice (probe.hip: k_hash_window): a probe sequence wraps inside its slice of mask + 1 sectors
    int32_t hash_mode = 0;         // kHashNone until built, then kHashK1 / kHashK3 / kHashTag
    bool accel_failed = false;     // a lookup structure could not be allocated (or tags collided): sorted search from now on
    std::mutex accel_mu;           // held by index_ensure_* from the "is it there" test to the recorded event: Joins of several
                                   // ctxs / threads (the slot workers of a general stream join) may ask for the same structure
    // Lookup structures are built on the INDEX's ctx (its stream, its pool: they live and die with the index); the
    // event is recorded behind the newest one, and a Join running on another ctx makes its stream wait for it.
    hipEvent_t accel_ready = nullptr;
    ~cph_index() {
        if (accel_ready) (void)hipEventDestroy(accel_ready);
    }
    int32_t sort_passes = 0;
    bool small_built = false;      // built by the one-launch path (small_build.hip)
    bool host_coded = false;       // built from codes the host formed (host_encode.hip: build_from_host_codes)
    uint64_t first_dup = UINT64_MAX;
    uint32_t* perm_host = nullptr; // pinned copy (lazy)
    size_t perm_host_cap = 0;
    cph::DevBuf first_dup_dev;     // u32 result of the adjacent-equal scan until it is read back
};

struct cph_chain_impl {
    cph_chain pub;                 // must stay first
    cph_ctx* ctx = nullptr;
    cph::DevBuf d_stream;
    cph::DevBuf d_rows[CPH_MAX_CHAIN];
    void* h_block = nullptr;       // pinned, out of the ctx's cache of pinned blocks (pinned_cache_get): a Join per batch of 8192 rows must not page-lock
    size_t h_cap = 0;
};

struct cph_matches_impl {
    cph_matches pub;               // must stay first: the public view
    cph_ctx* ctx = nullptr;
    cph::DevBuf d_lo, d_cnt, d_pidx, d_brow;
    void* h_block = nullptr;       // one pinned block holding the host copies (out of the ctx's cache of pinned blocks)
    size_t h_cap = 0;
};

// ---- cross-file entry points (host functions launching kernels) -------------------------------
namespace cph {

// A string column resident on the device.
struct DevCol {
    const uint8_t* data = nullptr;
    const void* offsets = nullptr;
    uint64_t nrows = 0;
    int32_t offset_bits = 32;
    uint32_t fixed_width = 0;   // > 0: every value has this many bytes, offsets unused (may be null)
    // segment of the values a key WINDOW looks at (keys longer than kMaxKeyBytes positions): bytes [skip, skip+take)
    // of every value.  take == 0xFFFFFFFF: to the end of the value (the column's last segment: a value longer than
    // the build side's longest stays longer than the window's maxlen and is recognised as absent).
    uint32_t skip = 0, take = 0xFFFFFFFFu;
    bool segmented() const { return skip != 0 || take != 0xFFFFFFFFu; }
    // virtual column of a split codec (CodecHost::split_col): split = 0x100 | delimiter byte, part 0 = the value up to and
    // including its first delimiter (all of it when it holds none), part 1 = what follows the first delimiter
    uint16_t split = 0;
    uint16_t part = 0;
};
// The key columns as the codec sees them: `real` = the table's (or the stream's) leading nreal key columns; a split
// codec gets its split column twice (prefix part, suffix part).  Returns the number of virtual columns written to out
// (room for kMaxKeyCols).
int codec_virtual_cols(const CodecHost& codec, const DevCol* real, int nreal, DevCol* out);
// Builds CodecHost::wide_disp / wide_slots from wdict; false when no perfect hash was found (two prefixes with one 32-bit
// hash: the caller then does without the split).
bool codec_wide_perfect_hash(CodecHost* codec);

// keycodec.hip
struct ColStats {              // per column, produced by one pass over the column
    uint32_t minlen, maxlen;
    uint32_t mask[kMaxKeyBytes][8];   // presence bitmap of byte values per position
};
Status codec_collect_stats(cph_ctx* ctx, const DevCol* cols, int32_t ncols, std::vector<ColStats>* out);
// the same in two halves, so that several indexes share one stream synchronisation (cph_index_build_many)
Status codec_stats_launch(cph_ctx* ctx, const DevCol* cols, int32_t ncols, DevBuf* dev_stats);
void codec_stats_finish(const DevCol* cols, int32_t ncols, const void* host_copy, std::vector<ColStats>* out);
Status codec_build(const std::vector<ColStats>& stats, CodecHost* codec);   // host only
// alphabets from a sample of the rows instead of a pass over all of them (keycodec.hip; capi.hip: BuildJob::sampled)
bool codec_sample_applies(const cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n);
size_t codec_sample_bytes();
Status codec_sample_launch(cph_ctx* ctx, const DevCol& col, uint64_t n, const void** host_copy);
void codec_sample_finish(const DevCol& col, const void* host_copy, std::vector<ColStats>* out);
bool codec_sample_checked(const CodecHost& codec, const DevCol* cols);
// When the per-position code needs several words: one more pass over the key columns collects the distinct
// joint symbols of every 7-position group; groups with few of them are dictionary-coded (codec rebuilt in place).
// On large inputs the dictionaries can be SPECULATIVE (spec != nullptr): they are taken from a sample of the rows, the
// encode kernel adds every window it does not find to the device sets and raises `miss`; the caller reads the flag
// with its next synchronisation and, when it is set, calls codec_groups_complete (the sets are complete by then: every
// row's windows were either found or inserted) and encodes again.  A sample that saw every window — the usual case for
// low-cardinality fields — saves the exact pass over all rows.
struct GroupChoice { int t, p0, span; double saved; uint32_t count; uint32_t tab; };   // tab: window descriptor (keycodec.hip)
struct GroupSpec {
    bool active = false;
    DevBuf slots, counts, miss;          // per table: hash set [kGroupSlots] u64, entry count u32; miss: rows with an unknown window (u32)
    std::vector<GroupChoice> chosen;     // the tables behind the codec's group heads
    CodecHost plain;                     // the codec without groups
};
Status codec_try_groups(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, CodecHost* codec, GroupSpec* spec = nullptr);
// missed = the encode kernel's count of rows with an unknown window; at or above kSpecGiveUp the kernel stopped
// early and the exact pass over all rows runs first.
constexpr uint32_t kSpecGiveUp = 1u << 16;
Status codec_groups_complete(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, uint32_t missed, GroupSpec* spec, CodecHost* codec);
Status codec_upload(cph_ctx* ctx, const CodecHost& codec, DevBuf* dev);
int codec_premultiplied_bits(const CodecHost& codec);   // 0 / 32 / 64
// Encodes the build-side keys.  key32: out32[n]; else out64[nwords][n].
// hist (optional): the sort's first-pass histogram request; `done` tells the caller whether the encode kernel
// produced it (only the single-column fast path does).
struct EncodeHist {
    uint32_t tile_rows = 0;       // keys per sort tile
    uint32_t digit_mask = 0;      // (1 << bits of the first digit) - 1
    uint32_t bins = 0;            // digits per pass of the sort (256 or 512 >= digit_mask + 1)
    uint32_t* counts = nullptr;   // device [bins][ntiles], digit-major
    bool done = false;
    // Direct sort of distinct keys over a FULL code space (radix_sort.hip): instead of writing the codes, the encode kernel stores
    // slot[code] = row straight away (slots: device u32[states], preset to 0xFFFFFFFF).  scattered = the kernel that ran did so
    // (only the single-column fast path can); otherwise the codes were written as usual.
    uint32_t* slots = nullptr;
    uint32_t slot_states = 0;
    bool scattered = false;
};
// cols = the TABLE's key columns (a split codec's virtual columns are formed inside).  miss (optional, device u32): set
// when a row did not code (split codecs only: see codec_try_split).
Status codec_encode_build(cph_ctx* ctx, const CodecHost& codec, const DevBuf& codec_dev, const DevCol* cols,
                          uint64_t n, void* out_codes, const EncodeHist* hist = nullptr, const GroupSpec* spec = nullptr,
                          uint32_t* miss = nullptr);
// The delimiter split (keycodec.hip "split codec"): tried when the plain code does not fit 32 bits.  stats: the plain
// statistics *codec was built from, or nullptr for ONE variable-length key column before any statistics exist (a large
// table then skips the plain statistics pass); codec->has_split() tells whether the split was taken.
Status codec_try_split(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, const std::vector<ColStats>* stats, CodecHost* codec,
                       bool speculate = false, bool* speculated = nullptr);
// The same for ONE variable-length column in host memory, its sample read by the host's threads (keycodec.hip; used by
// host_encode.hip: build_from_host_codes): always the speculative kind — the caller's encode loop checks every row.
Status codec_split_from_host(cph_ctx* ctx, const cph_host::HostCol& hc, uint64_t n, cph_host::BlockPool& pool, CodecHost* codec);
// Host-side encoding of literal values (cph_index_find).  Returns false when a
// value cannot occur in the index (symbol outside the alphabet / too long).
bool codec_encode_values_host(const CodecHost& codec, const cph_strval* values, int32_t nvalues,
                              uint64_t* q_exact, int32_t* nq, uint64_t* qlo, uint64_t* qhi);

// radix_sort.hip
// Stable LSD radix sort of (key,val) pairs over key bits [0,bits).  keys_in may be
// clobbered.  vals_in == nullptr means vals = 0..n-1.  On return *keys_out/*vals_out
// point at whichever of the two buffer pairs holds the result.
struct RadixPlan {
    int npass = 0, threads = 256, rbits = 8;
    uint32_t tile = 4096, ntiles = 0;
    int nb0 = 0;                  // bits of the first (least significant) digit
    size_t count_words() const { return ((size_t)1 << rbits) * ntiles; }
};
RadixPlan radix_plan(const cph_ctx* ctx, uint64_t n, int bits);
// counts (optional): a device buffer of plan.count_words() u32 the caller allocated; first_hist_done = it already
// holds the first pass's per-tile digit histogram (codec_encode_build produced it).
template <class K>
Status radix_sort_pairs(cph_ctx* ctx, K* keys_a, K* keys_b, uint32_t* vals_a, uint32_t* vals_b, bool vals_iota,
                        uint64_t n, int bits, K** keys_out, uint32_t** vals_out, int* passes,
                        uint32_t* counts = nullptr, bool first_hist_done = false);
// 32-bit codes WITH duplicates: MSD sort through counted LDS windows (counted_sort.hip); plan() says whether it applies
struct CountedSortPlan {
    uint32_t wbits = 0, k2 = 0, nb1 = 0, nwt = 0;
    bool two = false;
};
bool counted_sort_plan(const cph_ctx* ctx, uint64_t n, uint64_t states, CountedSortPlan* p, int max_wbits = 31, uint64_t min_rows = 1ull << 21);
// In two steps: begin (buffers, zeroed counters) | run.  (hist_done: somebody else filled the counters.  Counting the rows per window
// inside the split-codec encode kernel was tried in round 6: its LDS atomics and the 34 KB of counters cost the kernel 0.23 ms, the
// separate k_cs_hist pass 0.10 ms.)
struct CountedSort {
    CountedSortPlan p;
    DevBuf words, ent1, ent2;
    uint32_t* counts = nullptr;
    Status begin(cph_ctx* ctx, const CountedSortPlan& plan, uint64_t n);
    // the partition alone (probe.hip builds hash tables slice by slice from it): entries() grouped by window, window w at wbase()[w]
    Status partition(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* over_host, bool hist_done);
    const uint64_t* entries() const { return p.two ? ent2.as<uint64_t>() : ent1.as<uint64_t>(); }
    const uint32_t* flag() const { return counts + p.nwt; }
    const uint32_t* wbase() const { return counts + p.nwt + 1; }
    Status run(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out, uint32_t* first_dup_dev,
               uint32_t* over_host, bool hist_done);
};
Status counted_sort(cph_ctx* ctx, const CountedSortPlan& p, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out,
                    uint32_t* sorted_out, uint32_t* first_dup_dev, uint32_t* over_host);
void warm_counted_sort();
// distinct 32-bit codes over a dense space: one scatter instead of radix passes (optimistic; *flag raised on a duplicate)
Status direct_sort_distinct(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out,
                            uint32_t* flag, void* ranktab = nullptr, uint64_t rank_blocks = 0, bool* ranktab_written = nullptr);
// the same through LDS windows: a partition by the top code bits, then every window placed in LDS and streamed out (window_sort.hip)
struct ArithPlan;   // codec_device.hpp
Status direct_sort_windows_keys(cph_ctx* ctx, const uint64_t* keys, const ArithPlan& ap, uint64_t n, uint64_t states, uint32_t* perm_out,
                                uint32_t* sorted_out, uint32_t* flag, void* ranktab = nullptr, uint64_t rank_blocks = 0);
Status direct_sort_windows(cph_ctx* ctx, const uint32_t* codes, uint64_t n, uint64_t states, uint32_t* perm_out, uint32_t* sorted_out,
                           uint32_t* flag, void* ranktab = nullptr, uint64_t rank_blocks = 0);
// ... in steps, for a table whose codes arrive in chunks (host_encode.hip): begin | add(chunk) per chunk — the first partition level of
// that chunk, enqueued behind its upload — | finish.  Whether the direct sort applies (distinct keys expected, dense space) is the caller's call.
struct WindowSort {
    Status begin(cph_ctx* ctx, uint64_t n, uint64_t states);
    Status add(cph_ctx* ctx, const uint32_t* codes, uint64_t row0, uint64_t m, uint32_t* flag, const uint64_t* keys = nullptr, const ArithPlan* ap = nullptr);
    // ranktab != nullptr: the rank table of the index (uint2 {presence bits, keys before} per 32 codes, rank_blocks of them) is written too
    Status finish(cph_ctx* ctx, uint32_t* perm_out, uint32_t* sorted_out, uint32_t* flag, void* ranktab = nullptr, uint64_t rank_blocks = 0);
    ~WindowSort();
    WindowSort() = default;
    WindowSort(const WindowSort&) = delete;
    WindowSort& operator=(const WindowSort&) = delete;
    uint64_t n = 0, states = 0, nb1 = 0, nwin_total = 0;
    uint32_t nb2 = 1, shift1 = 0;
    bool two = false, started = false, finished = false;
    DevBuf ent1, ent2;
    DevBuf* words = nullptr;
    uint32_t *cur1 = nullptr, *cur2 = nullptr;
};
// the second half of it for a full code space whose slots the encode kernel already filled: every slot taken? + the sorted codes
Status direct_sort_finish_full(cph_ctx* ctx, const uint32_t* slots, uint64_t n, uint32_t* sorted_out, uint32_t* flag);
Status exclusive_scan_u32(cph_ctx* ctx, uint32_t* data, uint64_t n);
Status exclusive_scan_u32_total(cph_ctx* ctx, uint32_t* data, uint64_t n, uint32_t* total_out);   // total_out: device
Status exclusive_scan_u64(cph_ctx* ctx, uint64_t* data, uint64_t n, uint64_t* total_out);
Status gather_u64(cph_ctx* ctx, const uint64_t* src, const uint32_t* idx, uint64_t* dst, uint64_t n);
Status fill_iota_u32(cph_ctx* ctx, uint32_t* dst, uint64_t n);

// probe.hip
Status index_first_dup_launch(cph_ctx* ctx, cph_index* ix);
Status index_first_dup_read(cph_ctx* ctx, cph_index* ix);
void index_plan_table(cph_index* ix);                               // host decision only (table_entries)
// Lookup structures of Join, built on first use ON THE INDEX'S CTX (ix->ctx: its stream and pool); `ctx` is the
// caller's: when it is another one its stream is made to wait for the build.  An allocation failure is not an
// error: the index is marked (accel_failed) and the callers use the sorted search.
Status index_ensure_table(cph_ctx* ctx, const cph_index* ix);       // 8-byte entries {lo,row} / {lo,end}
Status index_ensure_rowtab(cph_ctx* ctx, const cph_index* ix);      // 4-byte build rows (duplicate-free indexes)
Status index_ensure_ranktab(cph_ctx* ctx, const cph_index* ix);     // presence bits + running count per 32 codes (duplicate-free indexes)
inline uint64_t ranktab_blocks(uint64_t table_entries) { return ((table_entries + 31) / 32 + 1) & ~1ull; }   // 32-code blocks, an even number of them
Status index_ensure_hash(cph_ctx* ctx, const cph_index* ix);        // hash table over the codes (hash_device.hpp)
bool index_wants_hash(const cph_index* ix);                         // no direct table planned and rows to look up
struct ProbeOut {
    DevBuf lo, cnt, pidx, brow;
    uint64_t nprobe = 0, nmatches = 0;
};
struct RowSel {                // optional selection of probe rows (device memory)
    const void* ptr = nullptr; // NULL: rows 0..nprobe-1
    int32_t bits = 32;         // 32 or 64
    uint64_t base = 0;         // subtracted from every entry
};
Status probe_run(cph_ctx* ctx, const cph_index* ix, const DevCol* cols, int32_t ncols, RowSel sel,
                 uint64_t nprobe, uint64_t probe_base, bool want_pairs, ProbeOut* out, bool positions = false);
Status index_find_device(cph_ctx* ctx, const cph_index* ix, const uint64_t* q_exact, int32_t nq, uint64_t qlo,
                         uint64_t qhi, uint64_t* lower, uint64_t* upper);
// nkeys query blocks of `stride` words each ([nq | exact words | qlo | qhi], nq = ~0: the key cannot occur, nq = 0: every row matches): one upload,
// one launch, one download
Status index_find_many_device(cph_ctx* ctx, const cph_index* ix, const uint64_t* queries, size_t stride, uint64_t nkeys,
                              uint64_t* lower, uint64_t* upper);

// index_ops.hip: an index as descriptor (host bytes) + sorted codes + perm (device arrays)
void index_desc_serialize(const cph_index* ix, std::vector<uint8_t>* out);
size_t index_desc_header_bytes();                                   // bytes index_desc_size needs to see
bool index_desc_size(const uint8_t* p, size_t n, size_t* need);     // size of the whole descriptor, from its header
bool index_desc_parse(const uint8_t* p, size_t n, cph_index* ix);
Status index_adopt_payload(cph_ctx* ctx, cph_index* ix);            // validate the device arrays, finish the index
size_t index_code_bytes(const cph_index* ix);                       // bytes of sorted codes per row

// chain.hip
struct ChainStep {
    const cph_index* index = nullptr;
    DevCol cols[kMaxKeyCols];
    int32_t ncols = 0;
    int32_t source = 0;        // cph_chain_step.source: 0 = cols belong to the stream table; k / -k = to the build table of step k-1
                               // (original row order / that index's sorted order)
};
struct ChainOut {
    DevBuf stream_row;
    DevBuf build_row[CPH_MAX_CHAIN];
    uint64_t nrows = 0;
    int32_t nsteps = 0;
    bool identity = false;     // stream_row[m] == probe_base + m for all m: stream_row is not materialised
};
// positions: build_row[k] holds the SORTED POSITION of the matching row in index k (an index into the index's sorted
// rows / its perm) instead of the original row id perm[position]
Status chain_run(cph_ctx* ctx, const ChainStep* steps, int nsteps, uint64_t probe_base, ChainOut* out, bool positions = false);
// asynchronous pieces of the fast path (stream_join.hip pipelines them over several streams)
bool chain_fast_path_ok(const ChainStep* steps, int nsteps);
Status chain_enqueue_dense(cph_ctx* ctx, const ChainStep* steps, int nsteps, uint64_t nprobe, uint64_t probe_base,
                           uint32_t* const* d_rows, uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total, bool positions = false);
// the same over host-formed key codes (host_encode.hip) that already sit in device memory
Status chain_enqueue_codes(cph_ctx* ctx, const cph_index* const* idx, const uint32_t* const* d_codes, int nsteps, uint64_t nprobe,
                           uint32_t* const* d_rows, uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total, bool positions);
uint64_t chain_dense_mask_words(uint64_t nprobe);
uint64_t chain_dense_count_words(uint64_t nprobe);

// every translation unit with kernels: forces its code object to load (called once per process from cph_ctx_create)
void warm_keycodec();
void warm_radix_sort();
void warm_probe();
void warm_chain();
void warm_materialize();
void warm_csv_ingest();
void warm_index_ops();
void warm_small_build();
void warm_window_sort();
// host_encode.hip: IndexOn over one short key column in host memory through host-formed codes; *taken = false: not applicable
// (or a row the sampled alphabets could not code / duplicates under the direct sort): the index is untouched, the general path runs
Status build_from_host_codes(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, cph_index* ix, bool unique, bool* taken);
void host_pool_destroy(cph_ctx* ctx);

// small_build.hip: IndexOn of a small table in one launch (one workgroup) and one synchronisation
constexpr int kSmallMaxPos = 64;              // byte positions of the key the one-workgroup build takes
enum : uint32_t { kSmallBuilt = 0, kSmallNotSmall = 1, kSmallPending = 0xFFFFFFFFu };
struct SmallResult {                          // written by k_small_build into pinned host memory
    uint32_t status;                          // kSmallBuilt / kSmallNotSmall (too many positions, code of several words)
    uint32_t first_dup;                       // first sorted position equal to its predecessor, 0xFFFFFFFF: none
    uint32_t bits, key32, passes;
    uint32_t minlen[kMaxKeyCols], maxlen[kMaxKeyCols];
    uint32_t mask[kSmallMaxPos][8];           // ColStats::mask of the positions, column-major
    uint64_t t[10];                           // wall_clock64() at the phase boundaries (ctx option codec_debug prints them)
    uint64_t codec_check;                     // small_codec_check over the kernel's own radices and weights: the host-rebuilt codec must give the same
};
// Fold of a codec's per-position radices and weights (device: what k_small_build encoded and sorted by; host: what codec_build
// rebuilt from the same statistics and what every later probe will encode by).
CPH_HD inline uint64_t small_codec_check(uint64_t h, uint64_t radix, uint64_t mult) {
    h = (h ^ radix) * 0x9E3779B97F4A7C15ull;
    return (h ^ mult) * 0xC2B2AE3D27D4EB4Full + 1;
}
struct SmallBufs {
    DevBuf ka, kb, va, vb, sorted, perm;
};
bool small_build_applies(const cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n);
Status small_build_launch(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, SmallBufs* bufs, SmallResult* res);
Status small_build_finish(cph_ctx* ctx, cph_index* ix, int32_t ncols, SmallBufs* bufs, const SmallResult* res, bool* not_small);

// capi.hip helpers
Status ensure_pinned_scratch(cph_ctx* ctx, size_t bytes);
Status pinned_cache_get(cph_ctx* ctx, size_t bytes, void** out, size_t* cap);   // a cached block of >= bytes, or a new one
void pinned_cache_put(cph_ctx* ctx, void* p, size_t cap);                        // back into the cache (at most 2 blocks kept)
Status pinned_upload(cph_ctx* ctx, size_t bytes, void** out);   // staging slot valid until the ring wraps
Status validate_cols(const cph_strcol* cols, int32_t ncols);
// Makes columns device resident (host columns are copied into pool blocks kept alive by `storage`).
Status stage_cols(cph_ctx* ctx, const cph_strcol* cols, int32_t ncols, std::vector<DevBuf>* storage, DevCol* out);

// Records the error text on the ctx and returns the status code (every extern "C" entry point ends through this).
inline int32_t fail_with(cph_ctx* ctx, const Status& s) {
    if (ctx) ctx->err = s.msg;
    return s.code;
}

// One value device -> host through the ctx's pinned scratch; synchronises the stream.
template <class T>
Status read_device_value(cph_ctx* ctx, const T* dev, T* host) {
    CPH_TRY(ensure_pinned_scratch(ctx, sizeof(T)));
    CPH_HIP_TRY(hipMemcpyAsync(ctx->pinned_scratch, dev, sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->pinned_scratch, sizeof(T));
    return {};
}

// Dynamic-LDS attribute + resident workgroups per CU of a kernel, set up once per (ctx, kernel, lds size).
Status kernel_setup(cph_ctx* ctx, const void* fn, int threads, size_t lds, int* blocks_per_cu);
Status device_cus(cph_ctx* ctx, int* cus);

// Grid for a grid-stride loop of 256-thread workgroups over n items.
inline unsigned grid_for_items(uint64_t n, unsigned cap = 8192) {
    uint64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    return (unsigned)(b ? b : 1);
}

// Times everything enqueued on ctx->stream during its lifetime when ctx->profiling is on
// (two HIP events on that stream); `bytes` = algorithmic bytes of the launch.
class ProfScope {
public:
    ProfScope(cph_ctx* ctx, const char* name, double bytes);
    ~ProfScope();
private:
    cph_ctx* ctx_;
    int idx_ = -1;
    hipEvent_t start_ = nullptr;
    double bytes_ = 0;
};

}  // namespace cph

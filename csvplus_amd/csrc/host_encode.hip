// host_encode.hip — key codes formed ON THE HOST, so that a stream that lives in host memory crosses PCIe as 4-byte codes
// instead of its key strings (BASELINE config 5; the Join stream of csvplus.go:553-556 as the cgo shim stages it).
//
// A stream row's key takes part in a Join only through its code under the index's codec (keycodec.hip): two keys are equal
// iff their codes are.  When that code is ONE word below 2^31 (decimal ids of any realistic table, short tags) the host can
// form it with the very LUT the device walks — 8-17 bytes of string per row and step become 4 — and
// cph_stream_join_submit_codes ships and joins the codes (chain.hip: k_chain_codes).  A key that cannot occur in the index
// (a byte outside an alphabet, a value longer than the longest index key) gets CPH_CODE_ABSENT and joins nothing, exactly
// like the device encode's invalid flag.
//
// The loops and the worker pool live in host_encode_kernels.hpp (no HIP types: tested on their own on CPU): 8-byte decimal ids
// take 1 ns per row and thread (AVX2: bytewise range check + pmaddubsw / pmaddwd), short variable-length ids 4 ns (one 8-byte
// load, positions unrolled), anything else the plain LUT walk; cph_host_encoder_run hands 65 536-row blocks of the chunk to
// the pool's workers and to the calling thread.
#include <chrono>
#include <new>

#include "codec_device.hpp"
#include "host_encode_kernels.hpp"

using namespace cph;

struct cph_host_encoder {
    std::vector<uint32_t> lutw;          // pre-multiplied LUT, [npos][257]; top bit: symbol outside the alphabet
    int32_t ncols = 0, npos = 0;
    int32_t col_start[kMaxKeyCols + 1] = {0};
    int32_t col_maxlen[kMaxKeyCols] = {0};
    bool arith = false, arith_vector = false;   // one column of 8-byte fixed-width values over contiguous alphabets: no table at all
    cph_host::Arith8 arith8{};
    std::unique_ptr<cph_host::BlockPool> pool;
    std::mutex run_mu;                   // one job at a time
};

static_assert(cph_host::kLutRow == kLutStride, "the host loops walk the device's LUT layout");
static_assert(cph_host::kCodeAbsent == CPH_CODE_ABSENT, "one ABSENT code");

// Rows [row0, row0 + n) of the columns -> out[0 .. n) on the pool's workers and the calling thread.  any_absent (optional) is
// raised when some row got CPH_CODE_ABSENT.
static void encode_rows_on(cph_host::BlockPool& pool, const cph_host_encoder& e, const cph_host::HostCol* hc, int ncols, uint64_t row0, uint64_t n,
                           uint32_t* out, std::atomic<uint32_t>* any_absent, bool nt = false) {
    uint32_t* base = out - row0;   // the loops index their output by row number
    auto note = [&](bool absent) {
        if (absent && any_absent) any_absent->store(1u, std::memory_order_relaxed);
    };
    if (e.arith && hc[0].fixed_width == 8) {
        const uint8_t* d = hc[0].data;
        if (e.arith_vector) pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_arith8_avx2(e.arith8, d, row0 + r0, row0 + r1, base, nt)); });
        else pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_arith8(e.arith8, d, row0 + r0, row0 + r1, base, nt)); });
    } else if (ncols == 1 && e.npos <= 8) {
        pool.run(n, [&](uint64_t r0, uint64_t r1) { note(cph_host::encode_lut_short(e.lutw.data(), e.npos, hc[0], row0 + r0, row0 + r1, base, nt)); });
    } else {
        pool.run(n, [&](uint64_t r0, uint64_t r1) {
            note(cph_host::encode_lut(e.lutw.data(), e.ncols, e.col_start, e.col_maxlen, hc, row0 + r0, row0 + r1, base, nt));
        });
    }
}

static void encoder_tables(const CodecHost& cd, cph_host_encoder* e);

extern "C" {

CPH_API int32_t cph_host_encoder_create(const cph_index* ix, int32_t nthreads, cph_host_encoder** out) {
    if (!ix || !out) return CPH_ERR_INVALID;
    *out = nullptr;
    cph_ctx* ctx = ix->ctx;
    const CodecHost& cd = ix->codec;
    if (!ix->windows.empty() || cd.nwords != 1 || cd.has_groups() || cd.npos < 1 || cd.word_states[0] > (1ull << 31))
        return fail_with(ctx, {CPH_ERR_INVALID, "cph_host_encoder_create: the keys of this index do not code in one word below 2^31 per position "
                                                "(dictionary / split codecs, long keys): ship the key strings (cph_stream_join_submit)"});
    auto* e = new (std::nothrow) cph_host_encoder();
    if (!e) return fail_with(ctx, {CPH_ERR_NOMEM, "out of host memory"});
    try {
        encoder_tables(cd, e);
        // workers: the loops are memory-bound well before every hardware thread is busy, and idle workers spin for a moment
        // before they sleep — half the hardware threads, at most 64 (+ the calling thread, which takes blocks too)
        int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency() / 2;
        if (nt < 1) nt = 1;
        if (nt > 256) nt = 256;
        if (nthreads <= 0 && nt > 64) nt = 64;
        e->pool.reset(new cph_host::BlockPool(nt - 1));
    } catch (const std::exception& ex) {
        delete e;
        return fail_with(ctx, {CPH_ERR_NOMEM, std::string("cph_host_encoder_create: ") + ex.what()});
    }
    *out = e;
    return CPH_OK;
}

}  // extern "C"

// The loops' tables from a codec whose code is one word below 2^31 (may throw std::bad_alloc).
static void encoder_tables(const CodecHost& cd, cph_host_encoder* e) {
    {
        e->ncols = cd.ncols;
        e->npos = cd.npos;
        for (int c = 0; c <= cd.ncols; c++) e->col_start[c] = cd.col_start[c];
        for (int c = 0; c < cd.ncols; c++) e->col_maxlen[c] = cd.col_maxlen[c];
        e->lutw.resize((size_t)cd.npos * kLutStride);
        for (size_t i = 0; i < e->lutw.size(); i++) {
            const uint16_t r = cd.lut[i];
            e->lutw[i] = r == kLutInvalid ? 0x80000000u : (uint32_t)((uint64_t)r * cd.mult[i / kLutStride]);
        }
        ArithPlan ap{};
        codec_arith_plan(cd, &ap);
        if (ap.enabled && ap.keylen == 8 && cd.ncols == 1) {
            e->arith = true;
            e->arith8.lo = (uint64_t)ap.lo[0] | ((uint64_t)ap.lo[1] << 32);
            e->arith8.rngc = (uint64_t)ap.rngc[0] | ((uint64_t)ap.rngc[1] << 32);
            for (int p = 0; p < 8; p++) {
                e->arith8.mult[p] = (uint32_t)cd.mult[(size_t)p];
                e->arith8.radix[p] = 0x80u - (uint32_t)((e->arith8.rngc >> (8 * p)) & 0xFFu);   // rngc byte = 0x7F - (radix - 1)
            }
            // The vector loop is 6x the scalar one on the build container's Xeon and 20x SLOWER on the GPU box's host
            // (profiles/r04_host_encode.txt: 0.92 against 18.5 G rows/s on 256 threads) — so it is not assumed, it is timed:
            // both loops over 32 768 valid keys, the faster one serves this encoder.
            if (cph_host::arith8_vector_ok(e->arith8)) {
                const uint64_t n = 32768;
                std::vector<uint8_t> keys(8 * n);
                for (uint64_t i = 0; i < 8 * n; i++) keys[i] = (uint8_t)(e->arith8.lo >> (8 * (i & 7)));
                std::vector<uint32_t> codes(n);
                auto time_of = [&](bool vec) {
                    double best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        const auto t0 = std::chrono::steady_clock::now();
                        if (vec) cph_host::encode_arith8_avx2(e->arith8, keys.data(), 0, n, codes.data());
                        else cph_host::encode_arith8(e->arith8, keys.data(), 0, n, codes.data());
                        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                    }
                    return best;
                };
                e->arith_vector = time_of(true) < time_of(false);
            }
        }
    }
}

extern "C" {

CPH_API int32_t cph_host_encoder_threads(const cph_host_encoder* e) { return e && e->pool ? e->pool->workers() + 1 : 0; }

// cols = the stream's key columns for the index (ALL its key columns, host memory), out_codes = nrows u32 (any host
// memory; pinned — cph_pinned_alloc — when cph_stream_join_submit_codes is to overlap its upload).  Blocks until done.
CPH_API int32_t cph_host_encoder_run(cph_host_encoder* e, const cph_strcol* cols, int32_t ncols, uint32_t* out_codes) {
    if (!e || !cols || !out_codes || ncols != e->ncols) return CPH_ERR_INVALID;
    const uint64_t n = cols[0].nrows;
    for (int c = 0; c < ncols; c++) {
        if (cols[c].mem != CPH_MEM_HOST || cols[c].nrows != n || (!cols[c].fixed_width && cols[c].offset_bits != 32 && cols[c].offset_bits != 64) ||
            (n && !cols[c].data && cols[c].fixed_width) || (!cols[c].fixed_width && !cols[c].offsets))
            return CPH_ERR_INVALID;
    }
    if (n == 0) return CPH_OK;
    cph_host::HostCol hc[kMaxKeyCols];
    for (int c = 0; c < ncols; c++) {
        hc[c].data = cols[c].data;
        hc[c].offsets = cols[c].offsets;
        hc[c].offset_bits = cols[c].offset_bits;
        hc[c].fixed_width = cols[c].fixed_width;
        // readable bytes = up to the end of the chunk's last value (the caller's buffer may end right there)
        hc[c].data_bytes = cols[c].fixed_width ? n * (uint64_t)cols[c].fixed_width : cph_host::col_offset(hc[c], n);
    }
    std::lock_guard<std::mutex> lk(e->run_mu);
    // a chunk's codes are read next by the DMA engine that uploads them (cph_stream_join_submit_codes), not by these cores: streaming
    // stores from 2^16 rows on (no read-for-ownership of the output lines — what made the build side's encode 2.7x faster in round 5,
    // profiles/r05_host_build.txt); a small batch, which its caller may well read back itself, keeps regular stores
    encode_rows_on(*e->pool, *e, hc, ncols, 0, n, out_codes, nullptr, n >= (1ull << 16));
    return CPH_OK;
}

CPH_API void cph_host_encoder_destroy(cph_host_encoder* e) { delete e; }

}  // extern "C"

// ---- IndexOn over a key column in HOST memory: the codes are formed on the host, only they cross PCIe ---------------------------
// createIndex (csvplus.go:707-738) as a cgo caller sees it hands over host columns.  The general path uploads the strings (8-22
// bytes per row), encodes on the device, sorts, and the caller then fetches perm: upload, build and download one after the other.
// For ONE key column of at most 8 byte positions the host does what the device's sample + encode kernels do:
//   1. alphabets from 65 536 rows spread over the table (as codec_sample_* does on the device) -> the codec (codec_build);
//   2. chunks of 2^22 rows are coded by the ctx's worker pool (host_encode_kernels.hpp: the SWAR range check + multiply-adds of
//      encode_arith8 for 8-byte decimal ids, one 8-byte load + unrolled LUT walk otherwise) into two pinned staging blocks and
//      uploaded (4 bytes per row) while the next chunk is coded; every row is checked against the sampled alphabets — a row
//      they cannot code (CPH_CODE_ABSENT) abandons this path and the general one (exact statistics) runs;
//   3. the device sorts the codes it received (the direct sort for UniqueIndexOn over a dense code space, else the radix passes
//      + the adjacent-equal scan), exactly as behind its own encode kernel.
// The index is the one the general path builds: same codec (the sampled alphabets are the exact ones when no row missed), same
// stable order.
namespace cph {

struct HostPool {
    cph_host::BlockPool pool;
    int bound_node = -2;   // NUMA node the workers are bound to (-2: never bound, -1: unbound again)
    explicit HostPool(int workers) : pool(workers) {}
};

// ---- NUMA placement of the wide pool ----------------------------------------------------------------------------------------
// A two-socket host: 96 threads land on both sockets, and the ones on the far socket read the column through the inter-socket link —
// the same build then takes 19 ms or 95 ms depending on where the scheduler put them (profiles/r06_host_split.txt).  The workers
// are bound to the CPUs of the node that holds the column's first page (get_mempolicy; where the syscall is not permitted: the node
// the calling thread runs on, which is where a first-touch allocation of the caller sits).
#if defined(__linux__)
}  // namespace cph
#include <sys/syscall.h>
#include <unistd.h>
namespace cph {
static int numa_node_of_addr(const void* addr) {
    int node = -1;
    const long rc = syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void*>(addr), 3ul /* MPOL_F_NODE | MPOL_F_ADDR */);
    return rc == 0 ? node : -1;
}
static bool numa_node_cpus(int node, cpu_set_t* set) {   // /sys/devices/system/node/nodeN/cpulist: "0-63,128-191"
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    CPU_ZERO(set);
    int count = 0;
    for (const char* p = buf; *p;) {
        if (*p < '0' || *p > '9') { p++; continue; }
        char* e;
        long a = strtol(p, &e, 10), b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); count++; }
        p = e;
    }
    return count > 0;
}
static int numa_node_of_cpu(int cpu) {
    for (int node = 0; node < 64; node++) {
        cpu_set_t set;
        if (!numa_node_cpus(node, &set)) { if (node > 8) break; else continue; }
        if (CPU_ISSET(cpu, &set)) return node;
    }
    return -1;
}
static void bind_pool_to_data(cph_ctx* ctx, HostPool* hp, const void* data) {
    if (!ctx->host_numa) {
        if (hp->bound_node >= 0) {   // switched off after a bound run: every CPU again
            cpu_set_t all;
            CPU_ZERO(&all);
            for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &all);
            hp->pool.set_affinity(all);
            hp->bound_node = -1;
        }
        return;
    }
    int node = numa_node_of_addr(data);
    if (node < 0) node = numa_node_of_cpu(sched_getcpu());
    if (node < 0 || node == hp->bound_node) return;
    cpu_set_t set;
    if (!numa_node_cpus(node, &set)) return;
    hp->pool.set_affinity(set);
    hp->bound_node = node;
    if (ctx->codec_debug) fprintf(stderr, "[cph] host pool of %d workers bound to NUMA node %d (%d CPUs)\n", hp->pool.workers(), node, CPU_COUNT(&set));
}
#else
static void bind_pool_to_data(cph_ctx*, HostPool*, const void*) {}
#endif

// CPUs the process may use per scheduling period: the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1),
// 1e9 when there is none.  A container with 256 visible hardware threads and a quota of 16 runs 96 busy threads for a sixth of each
// 100 ms period and is THROTTLED for the rest: the split loop's 2 CPU-seconds per 1e8 rows then take 95-180 ms instead of 20
// (profiles/r06_host_split.txt) — the pool sizes and the choice below go by the quota, not by the thread count.
static double cgroup_cpu_quota() {
    auto read2 = [](const char* path, char* buf, size_t cap) -> bool {
        FILE* f = fopen(path, "r");
        if (!f) return false;
        const size_t got = fread(buf, 1, cap - 1, f);
        fclose(f);
        buf[got] = 0;
        return got > 0;
    };
    char a[128], b[128];
    if (read2("/sys/fs/cgroup/cpu.max", a, sizeof a)) {
        if (strncmp(a, "max", 3) == 0) return 1e9;
        double quota = 0, period = 0;
        if (sscanf(a, "%lf %lf", &quota, &period) == 2 && quota > 0 && period > 0) return quota / period;
        return 1e9;
    }
    if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", a, sizeof a) && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", b, sizeof b)) {
        const double quota = atof(a), period = atof(b);
        if (quota > 0 && period > 0) return quota / period;
    }
    return 1e9;
}
static cph_host::BlockPool* ctx_host_pool(cph_ctx* ctx) {
    if (!ctx->host_pool) {
        int nt = ctx->host_threads > 0 ? ctx->host_threads : (int)std::thread::hardware_concurrency() / 2;
        if (nt < 1) nt = 1;
        if (nt > 256) nt = 256;
        if (ctx->host_threads <= 0 && nt > 32) nt = 32;   // memory-bound on the NUMA node of the pinned buffers well before that (profiles/r05_host_build.txt)
        if (ctx->host_threads <= 0 && (double)nt > cgroup_cpu_quota()) nt = (int)cgroup_cpu_quota() < 1 ? 1 : (int)cgroup_cpu_quota();   // (busy threads beyond the quota get the whole cgroup throttled)
        try {
            ctx->host_pool = new HostPool(nt - 1);
        } catch (const std::exception&) {
            return nullptr;
        }
    }
    return &static_cast<HostPool*>(ctx->host_pool)->pool;
}
// threads a compute-bound host loop should use: 3/8 of the hardware threads, at most 96, at most the CPU quota
static int wide_pool_threads(const cph_ctx* ctx) {
    if (ctx->host_split_threads > 0) return ctx->host_split_threads > 256 ? 256 : ctx->host_split_threads;
    int nt = (int)std::thread::hardware_concurrency() * 3 / 8;
    if (nt > 96) nt = 96;
    const double q = cgroup_cpu_quota();
    if ((double)nt > q) nt = (int)q;
    return nt < 1 ? 1 : nt;
}
// the pool of the compute-bound loops (the split codec's; ctx option host_split_threads)
static cph_host::BlockPool* ctx_host_pool_wide(cph_ctx* ctx) {
    if (!ctx->host_pool_wide) {
        const int nt = wide_pool_threads(ctx);
        try {
            ctx->host_pool_wide = new HostPool(nt - 1);
        } catch (const std::exception&) {
            return nullptr;
        }
    }
    return &static_cast<HostPool*>(ctx->host_pool_wide)->pool;
}
void host_pool_destroy(cph_ctx* ctx) {
    delete static_cast<HostPool*>(ctx->host_pool);
    ctx->host_pool = nullptr;
    delete static_cast<HostPool*>(ctx->host_pool_wide);
    ctx->host_pool_wide = nullptr;
}

// per-position byte presence, shortest / longest value of rows 0, step, 2 step, ...; false: a value beyond 8 bytes.  The rows are
// spread over the whole table (one cache miss each): the pool's threads share them.
static bool host_sample(cph_host::BlockPool& pool, const cph_host::HostCol& c, uint64_t n, ColStats* st) {
    const uint64_t step = n >> 16 ? n >> 16 : 1;
    const uint64_t nsel = (n + step - 1) / step;
    memset(st, 0, sizeof *st);
    std::mutex mu;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    bool too_long = false;
    pool.run(nsel, [&](uint64_t i0, uint64_t i1) {
        uint32_t mask[8][8] = {{0}};
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        bool bad = false;
        for (uint64_t i = i0; i < i1 && !bad; i++) {
            const uint64_t r = i * step;
            uint64_t b, l;
            if (c.fixed_width) {
                b = r * (uint64_t)c.fixed_width;
                l = c.fixed_width;
            } else {
                b = cph_host::col_offset(c, r);
                l = cph_host::col_offset(c, r + 1) - b;
            }
            if (l > 8) { bad = true; break; }
            lo = (uint32_t)l < lo ? (uint32_t)l : lo;
            hi = (uint32_t)l > hi ? (uint32_t)l : hi;
            for (uint64_t q = 0; q < l; q++) {
                const uint8_t v = c.data[b + q];
                mask[q][v >> 5] |= 1u << (v & 31);
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        too_long |= bad;
        mn = lo < mn ? lo : mn;
        mx = hi > mx ? hi : mx;
        for (int q = 0; q < 8; q++)
            for (int w = 0; w < 8; w++) st->mask[q][w] |= mask[q][w];
    }, 1024);
    st->minlen = mn;
    st->maxlen = mx;
    return !too_long;
}

Status build_from_host_codes(cph_ctx* ctx, const cph_strcol* keycols, int32_t nkeycols, cph_index* ix, bool unique, bool* taken) {
    *taken = false;
    if (!ctx->host_build || nkeycols != 1) return {};
    const cph_strcol& kc = keycols[0];
    const uint64_t n = kc.nrows;
    if (kc.mem != CPH_MEM_HOST || n < (1ull << 20) || n >= 0xFFFFFFFFull) return {};
    if (kc.fixed_width ? (kc.fixed_width > 8 || !kc.data) : (!kc.offsets || (kc.offset_bits != 32 && kc.offset_bits != 64))) return {};
    cph_host::HostCol hc;
    hc.data = kc.data;
    hc.offsets = kc.offsets;
    hc.offset_bits = kc.offset_bits;
    hc.fixed_width = kc.fixed_width;
    hc.data_bytes = kc.fixed_width ? n * (uint64_t)kc.fixed_width : cph_host::col_offset(hc, n);
    if (!kc.fixed_width && !kc.data && hc.data_bytes) return {};
    using clk = std::chrono::steady_clock;
    const auto t_enter = clk::now();
    cph_host::BlockPool* pool = ctx_host_pool(ctx);
    if (!pool) return {};
    // the codec: the per-position code of short keys (<= 8 bytes, one word below 2^31), else — variable-length values — the split codec
    // of keycodec.hip from the same sample the device would take (prefix dictionary + suffix positions: config 3's 10-22 byte keys)
    CodecHost cd;
    bool plain = false;
    {
        std::vector<ColStats> stats(1);
        plain = host_sample(*pool, hc, n, &stats[0]) && stats[0].maxlen != 0;
        if (plain && !kc.fixed_width) {
            // variable-length values: the shortest and the longest value EXACTLY (one pass over the offsets, 4-8 bytes per row) — a table
            // of decimal ids holds ten one-digit values in a hundred million, which no sample shows, and the pad symbol of every
            // position behind the shortest value belongs to the alphabets
            std::atomic<uint32_t> mn{stats[0].minlen}, mx{stats[0].maxlen};
            pool->run(n, [&](uint64_t r0, uint64_t r1) {
                uint64_t lo = ~0ull, hi = 0, b = cph_host::col_offset(hc, r0);
                for (uint64_t r = r0; r < r1; r++) {
                    const uint64_t e = cph_host::col_offset(hc, r + 1), l = e - b;
                    lo = l < lo ? l : lo;
                    hi = l > hi ? l : hi;
                    b = e;
                }
                const uint32_t lo32 = lo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lo, hi32 = hi > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)hi;
                uint32_t cur = mn.load(std::memory_order_relaxed);
                while (lo32 < cur && !mn.compare_exchange_weak(cur, lo32, std::memory_order_relaxed)) {}
                cur = mx.load(std::memory_order_relaxed);
                while (hi32 > cur && !mx.compare_exchange_weak(cur, hi32, std::memory_order_relaxed)) {}
            });
            if (mx.load() > 8 || mx.load() != stats[0].maxlen) plain = false;   // (a longer value than any sampled one: its bytes are unknown)
            stats[0].minlen = mn.load();
        }
        if (plain) {
            CPH_TRY(codec_build(stats, &cd));
            if (cd.nwords != 1 || !cd.key32 || cd.has_groups() || cd.has_split() || cd.npos < 1 || cd.word_states[0] > (1ull << 31)) plain = false;
        }
    }
    const bool split = !plain;
    if (split) {
        if (kc.fixed_width || ctx->host_split == 0) return {};
        if (ctx->host_split == 1) {
            // worth it?  The loop costs ~12 ns per row and thread (measured: 1e8 rows in 72 ms on the 16 threads a quota of 16 CPUs
            // allows, 19-23 ms on 96 threads in the periods the quota did not throttle); the strings cross PCIe at ~50 GB/s: 44 ms.
            const double encode_ms = (double)n * 12e-6 / (double)wide_pool_threads(ctx);
            const double upload_ms = ((double)hc.data_bytes + (double)n * (kc.offset_bits / 8)) / 50e6;
            if (encode_ms >= upload_ms) return {};
        }
        cd = CodecHost{};
        cph_host::BlockPool* wide = ctx_host_pool_wide(ctx);
        if (wide) bind_pool_to_data(ctx, static_cast<HostPool*>(ctx->host_pool_wide), hc.data);
        CPH_TRY(codec_split_from_host(ctx, hc, n, wide ? *wide : *pool, &cd));
        if (!cd.has_split()) return {};
    }
    const auto t_codec = clk::now();
    cph_host_encoder enc;
    cph_host::SplitEnc<WideKey> se{};
    std::vector<uint32_t> split_lutw;
    try {
        if (!split) {
            encoder_tables(cd, &enc);
        } else {
            const int p0 = cd.col_start[0], ps = cd.col_start[1];
            se.delim = cd.split_byte;
            se.vmax = (uint32_t)cd.split_maxlen;
            se.smaxlen = (uint32_t)cd.col_maxlen[1];
            se.pmult = (uint32_t)cd.mult[(size_t)p0];
            se.hmask = (uint32_t)cd.wide_slots.size() - 1u;
            se.dmask = (uint32_t)cd.wide_disp.size() - 1u;
            se.disp = cd.wide_disp.data();
            se.slots = cd.wide_slots.data();
            se.dict = cd.wdict.data();
            split_lutw.resize((size_t)se.smaxlen * kLutStride);
            for (size_t i = 0; i < split_lutw.size(); i++) {
                const uint16_t r = cd.lut[(size_t)ps * kLutStride + i];
                split_lutw[i] = r == kLutInvalid ? 0x80000000u : (uint32_t)((uint64_t)r * cd.mult[(size_t)ps + i / kLutStride]);
            }
            se.lutw = split_lutw.data();
        }
    } catch (const std::exception&) {
        return {};
    }
    // the split loop computes (a hash, a dictionary entry, 5-16 table loads per row): it scales with the threads where the 8-byte loops
    // wait for DRAM — its own, wider pool
    cph_host::BlockPool* epool = pool;
    if (split) {
        epool = ctx_host_pool_wide(ctx);
        if (!epool) epool = pool;
    }

    // ---- code + upload, chunk by chunk ----
    constexpr uint64_t kChunk = 1ull << 22;
    DevBuf ka, va;
    CPH_TRY(ka.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(va.alloc(&ctx->pool, n * sizeof(uint32_t)));
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t* part_done_p = nullptr;
    auto cleanup = [&]() {
        if (part_done_p && *part_done_p) (void)hipEventDestroy(*part_done_p);
        for (int k = 0; k < 2; k++) {
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (stage[k]) pinned_cache_put(ctx, stage[k], stage_cap[k]);
        }
    };
    Status st;
    for (int k = 0; k < 2 && st.ok(); k++) {
        st = pinned_cache_get(ctx, kChunk * sizeof(uint32_t), &stage[k], &stage_cap[k]);
        if (st.ok() && hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) st = {CPH_ERR_HIP, "hipEventCreate failed"};
    }
    // UniqueIndexOn over a dense code space: the direct sort (window_sort.hip), its first partition level chunk by chunk behind the uploads
    const uint64_t states = cd.word_states[0];
    const bool direct = unique && ctx->direct_sort == 1 && n >= (1ull << 16) && states >= n && states <= 2 * n && states < 0xFFFFFFFFull;
    WindowSort ws;
    uint32_t* miss = nullptr;
    hipEvent_t part_done = nullptr;
    if (direct && st.ok()) {
        miss = host_word(ctx);
        if (!miss) st = {CPH_ERR_HIP, "no pinned host memory for the report words of a build"};
        if (st.ok()) st = ws.begin(ctx, n, states);
        // the partition kernels run on the ctx's SECOND stream, each behind its chunk's copy (an event): on the copies' own stream
        // the next upload would wait for them
        if (st.ok() && !ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            ctx->side_stream = nullptr;
        }
        if (st.ok() && ctx->side_stream && hipEventCreateWithFlags(&part_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            part_done = nullptr;
        }
    }
    const bool part_side = direct && ctx->side_stream && part_done;
    part_done_p = &part_done;
    std::atomic<uint32_t> absent{0};
    uint64_t nchunks = 0;
    const auto t_begin = clk::now();
    double t_wait = 0, t_enc = 0;
    for (uint64_t r0 = 0; r0 < n && st.ok(); r0 += kChunk, nchunks++) {
        const int slot = (int)(nchunks & 1);
        const uint64_t m = n - r0 < kChunk ? n - r0 : kChunk;
        const auto t0 = clk::now();
        if (nchunks >= 2 && hipEventSynchronize(ev[slot]) != hipSuccess) { st = {CPH_ERR_HIP, "hipEventSynchronize failed"}; break; }
        const auto t1 = clk::now();
        if (!split) {
            encode_rows_on(*pool, enc, &hc, 1, r0, m, static_cast<uint32_t*>(stage[slot]), &absent, /*nt=*/true);
        } else {
            uint32_t* base = static_cast<uint32_t*>(stage[slot]) - r0;   // the loop indexes its output by row number
            epool->run(m, [&](uint64_t a, uint64_t b) {
                if (cph_host::encode_split(se, hc, r0 + a, r0 + b, base, /*nt=*/true,
                                           [](uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) { return wide_hash_lo(w0, w1, w2, w3, len); }))
                    absent.store(1u, std::memory_order_relaxed);
            }, 8192);
        }
        t_wait += std::chrono::duration<double>(t1 - t0).count();
        t_enc += std::chrono::duration<double>(clk::now() - t1).count();
        if (absent.load(std::memory_order_relaxed)) break;   // a row the sampled alphabets cannot code: the exact path
        if (hipMemcpyAsync(ka.as<uint32_t>() + r0, stage[slot], m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ev[slot], ctx->stream) != hipSuccess)
            st = {CPH_ERR_HIP, "upload of host-formed codes failed"};
        if (direct && st.ok()) {
            if (part_side) {
                if (hipStreamWaitEvent(ctx->side_stream, ev[slot], 0) != hipSuccess) st = {CPH_ERR_HIP, "hipStreamWaitEvent failed"};
                SideStream on_side(ctx, true);
                if (st.ok()) st = ws.add(ctx, ka.as<uint32_t>() + r0, r0, m, miss);
            } else {
                st = ws.add(ctx, ka.as<uint32_t>() + r0, r0, m, miss);
            }
        }
    }
    if (part_side) {   // the rest of the sort (ctx->stream) behind the last partition kernel
        if (st.ok() && (hipEventRecord(part_done, ctx->side_stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, part_done, 0) != hipSuccess))
            st = {CPH_ERR_HIP, "joining the partition stream failed"};
    }
    if (!st.ok() || absent.load()) {
        (void)hipStreamSynchronize(ctx->stream);
        if (part_side) (void)hipStreamSynchronize(ctx->side_stream);
        cleanup();
        return st;   // (ok + !taken: the caller runs the general path)
    }

    // ---- the index around the codes ----
    ix->ctx = ctx;
    ix->nrows = n;
    ix->table_rows = n;
    ix->nkeycols = 1;
    ix->codec = cd;
    ix->host_coded = true;
    auto run = [&]() -> Status {
        CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
        if (direct) {
            CPH_TRY(ws.finish(ctx, va.as<uint32_t>(), ka.as<uint32_t>(), miss));
            CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (*(volatile uint32_t*)miss) return {CPH_ERR_DUPLICATE, "#direct-duplicates"};   // (caught below: the general path finds WHERE)
            ix->sorted_codes = std::move(ka);
            ix->perm = std::move(va);
            ix->sort_passes = 0;
            ix->first_dup = UINT64_MAX;
        } else {
            DevBuf kb, vb;
            CPH_TRY(kb.alloc(&ctx->pool, n * sizeof(uint32_t)));
            // duplicates allowed, a window of the code space holds a few thousand rows: the counted LDS windows (counted_sort.hip)
            CountedSortPlan csp;
            if (ctx->counted_sort && counted_sort_plan(ctx, n, states, &csp)) {
                uint32_t* over = host_word(ctx);
                if (!over) return {CPH_ERR_HIP, "no pinned host memory for the report words of a build"};
                CPH_TRY(ix->first_dup_dev.alloc(&ctx->pool, sizeof(uint32_t)));
                CPH_TRY(counted_sort(ctx, csp, ka.as<uint32_t>(), n, states, va.as<uint32_t>(), kb.as<uint32_t>(), ix->first_dup_dev.as<uint32_t>(), over));
                CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
                if (*(volatile uint32_t*)over == 0) {
                    ix->sorted_codes = std::move(kb);
                    ix->perm = std::move(va);
                    ix->sort_passes = 0;
                    CPH_TRY(index_first_dup_read(ctx, ix));
                    index_plan_table(ix);
                    return {};
                }
                ix->first_dup_dev.reset();   // a window beyond its capacity: the classic passes over the same codes
            }
            CPH_TRY(vb.alloc(&ctx->pool, n * sizeof(uint32_t)));
            uint32_t *kout, *vout;
            int passes = 0;
            CPH_TRY(radix_sort_pairs<uint32_t>(ctx, ka.as<uint32_t>(), kb.as<uint32_t>(), va.as<uint32_t>(), vb.as<uint32_t>(), true, n,
                                               cd.word_bits[0], &kout, &vout, &passes));
            ix->sorted_codes = std::move(kout == ka.as<uint32_t>() ? ka : kb);
            ix->perm = std::move(vout == va.as<uint32_t>() ? va : vb);
            ix->sort_passes = passes;
            CPH_TRY(index_first_dup_launch(ctx, ix));
            CPH_TRY(index_first_dup_read(ctx, ix));
        }
        index_plan_table(ix);
        return {};
    };
    const auto t_loop = clk::now();
    st = run();
    if (ctx->codec_debug)
        fprintf(stderr, "[cph] host-coded build: %llu rows, %llu chunks, %d threads, split=%d arith=%d vector=%d: sample + codec %.2f ms, tables + buffers %.2f ms, encode %.2f ms, waits for staging slots %.2f ms, loop %.2f ms, "
                        "upload tail + sort + wait %.2f ms\n", (unsigned long long)n, (unsigned long long)nchunks, epool->workers() + 1, (int)split, (int)enc.arith, (int)enc.arith_vector,
                std::chrono::duration<double>(t_codec - t_enter).count() * 1e3, std::chrono::duration<double>(t_begin - t_codec).count() * 1e3, t_enc * 1e3, t_wait * 1e3, std::chrono::duration<double>(t_loop - t_begin).count() * 1e3,
                std::chrono::duration<double>(clk::now() - t_loop).count() * 1e3);
    cleanup();
    if (!st.ok()) {
        (void)hipStreamSynchronize(ctx->stream);
        const bool dup = st.code == CPH_ERR_DUPLICATE && st.msg == "#direct-duplicates";
        // leave the index as it came: the general path fills it
        ix->codec = CodecHost{};
        ix->codec_dev.reset(); ix->sorted_codes.reset(); ix->perm.reset(); ix->first_dup_dev.reset();
        ix->host_coded = false;
        ix->first_dup = UINT64_MAX;
        return dup ? Status{} : st;
    }
    *taken = true;
    return {};
}

}  // namespace cph

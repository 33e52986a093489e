// small_build.hip — IndexOn for SMALL tables in ONE launch and ONE synchronisation.
//
// The reference's own benchmarks build indexes over 120 and 10 000 rows (csvplus_test.go:1052-1102); through the general
// path (keycodec.hip + radix_sort.hip: statistics | host codec | encode, 3-5 sort passes of 5 launches, adjacent-equal
// scan = ~25 launches and two host round trips) such a build costs ~0.1 ms whatever its size.  Here one workgroup does
// all of createIndex (csvplus.go:707-738) for up to `small_build_rows` rows:
//   0  value lengths per key column (min / max)                       -> byte positions of the key
//   1  byte-presence flags per position in LDS                        -> ColStats (what k_col_stats produces)
//   2  the per-position rank LUT and the mixed-radix weights, built ON THE DEVICE exactly as codec_build /
//      codec_split_words build them on the host (keycodec.hip) — the host rebuilds the very same codec from the
//      statistics after the one synchronisation, for later probes
//   3  encode every row (LUT in LDS)
//   4  stable LSD radix sort of (code, row) in global scratch (L2 resident), 8-bit digits, ranks by wave-ballot digit
//      matching like k_radix_scatter: wave w owns a contiguous chunk of the rows, so (wave, iteration, lane) order is
//      input order
//   5  sorted codes + permutation out, first adjacent-equal position (sort.Sort + the unique check, csvplus.go:736, :716-726)
// The result block (status, statistics, first duplicate) is written straight into pinned host memory.
// Keys the single-word per-position code cannot take (more than kSmallMaxPos byte positions, more than 2^63 states)
// report kSmallNotSmall and go through the general path.  Integer / byte work on one CU; no MFMA.
#include "codec_device.hpp"

namespace cph {

constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / kWave;
constexpr int kSmallItems = 16;   // keys per lane the sort holds in registers: kSmallWaves * 64 * kSmallItems = 16384 rows at most
constexpr int kSmallMaxRows = kSmallWaves * kWave * kSmallItems;
static_assert(kSmallMaxPos * 256 == kSmallWaves * 256 * (int)sizeof(uint32_t), "flags and histograms share one LDS block");

struct SmallArg {
    ColsArg cols;
    int32_t ncols;
    uint32_t n;
    uint64_t* ka;
    uint64_t* kb;
    uint32_t* va;
    uint32_t* vb;
    void* sorted;       // u32[n] (key32) or u64[n]
    uint32_t* perm;
    SmallResult* res;   // pinned host memory
};

__device__ __forceinline__ uint32_t small_bits_needed(uint64_t states) {   // bits_needed (keycodec.hip)
    return states <= 1 ? 0u : 64u - (uint32_t)__builtin_clzll(states - 1);
}

constexpr int kSmallRows = 8;   // rows a thread walks together: their loads are issued before any of them is waited for

// [begin, begin + len) of kSmallRows rows of one (unsegmented) column, all offset loads in flight together
__device__ __forceinline__ void small_spans(const DevCol& col, const uint32_t (&row)[kSmallRows], uint64_t (&b)[kSmallRows],
                                            uint32_t (&l)[kSmallRows]) {
    if (col.fixed_width) {
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) { b[k] = (uint64_t)row[k] * col.fixed_width; l[k] = col.fixed_width; }
    } else if (col.offset_bits == 32) {
        const uint32_t* off = reinterpret_cast<const uint32_t*>(col.offsets);
        uint32_t x[kSmallRows], y[kSmallRows];
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) { x[k] = off[row[k]]; y[k] = off[row[k] + 1]; }
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) { b[k] = x[k]; l[k] = y[k] - x[k]; }
    } else {
        const uint64_t* off = reinterpret_cast<const uint64_t*>(col.offsets);
        uint64_t x[kSmallRows], y[kSmallRows];
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) { x[k] = off[row[k]]; y[k] = off[row[k] + 1]; }
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) {
            b[k] = x[k];
            const uint64_t d = y[k] - x[k];
            l[k] = d > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d;
        }
    }
}

// lanes of the wave (valid ones only) whose digit of NB bits equals the caller's
template <int NB>
__device__ __forceinline__ uint64_t small_peers(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}
__device__ __forceinline__ uint64_t small_peers_nb(uint32_t d, bool valid, int nb) {   // nb is workgroup-uniform
    switch (nb) {
        case 1: return small_peers<1>(d, valid);
        case 2: return small_peers<2>(d, valid);
        case 3: return small_peers<3>(d, valid);
        case 4: return small_peers<4>(d, valid);
        case 5: return small_peers<5>(d, valid);
        case 6: return small_peers<6>(d, valid);
        case 7: return small_peers<7>(d, valid);
        default: return small_peers<8>(d, valid);
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_small_build(const SmallArg a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_block[kSmallMaxPos * 256];   // phase 1: flags; phase 4: histograms
    __shared__ uint16_t s_lut[kSmallMaxPos * kLutStride];
    __shared__ uint64_t s_mult[kSmallMaxPos];
    __shared__ uint16_t s_radix[kSmallMaxPos];
    __shared__ uint32_t s_min[kMaxKeyCols], s_max[kMaxKeyCols], s_start[kMaxKeyCols + 1];
    __shared__ uint32_t s_scan[kSmallWaves + 1];
    __shared__ uint32_t s_mask[kSmallMaxPos * 8];
    __shared__ uint64_t s_t[10];
    __shared__ uint32_t s_status, s_bits, s_key32, s_dup;

    const int tid = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t n = a.n;
    const int ncols = a.ncols;

    if (tid == 0) s_t[0] = wall_clock64();
    // ---- 0: value lengths ----
    if (tid < kMaxKeyCols) { s_min[tid] = 0xFFFFFFFFu; s_max[tid] = 0; }
    if (tid == 0) { s_status = kSmallBuilt; s_dup = 0xFFFFFFFFu; }
    __syncthreads();
    // EVERY thread walks the same number of batches (rows past the end repeat the last row: harmless for statistics), so
    // that the wave reductions below see defined values in all 64 lanes
    for (uint32_t base = 0; base < n; base += kSmallThreads * kSmallRows) {
        const uint32_t r0 = base + (uint32_t)tid;
        uint32_t row[kSmallRows];
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) {
            const uint32_t r = r0 + (uint32_t)k * kSmallThreads;
            row[k] = r < n ? r : n - 1;
        }
        for (int c = 0; c < ncols; c++) {
            uint64_t b[kSmallRows];
            uint32_t l[kSmallRows];
            small_spans(a.cols.c[c], row, b, l);
            uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
            for (int k = 0; k < kSmallRows; k++) {
                mn = l[k] < mn ? l[k] : mn;
                mx = l[k] > mx ? l[k] : mx;
            }
            mn = wave_min(mn);
            mx = wave_max(mx);
            if (lane == 0) { atomicMin(&s_min[c], mn); atomicMax(&s_max[c], mx); }
        }
    }
    lds_atomics_barrier();
    if (tid == 0) {
        uint64_t start = 0;
        for (int c = 0; c < ncols; c++) {
            s_start[c] = (uint32_t)start;
            start += s_max[c];
            if (start > (uint64_t)kSmallMaxPos) { s_status = kSmallNotSmall; break; }
        }
        s_start[ncols] = (uint32_t)start;
    }
    __syncthreads();
    if (s_status != kSmallBuilt) {
        if (tid == 0) { a.res->status = kSmallNotSmall; __threadfence_system(); }
        return;
    }
    const int npos = (int)s_start[ncols];

    if (tid == 0) s_t[1] = wall_clock64();
    // ---- 1: byte presence per position ----
    {
        uint4* z = reinterpret_cast<uint4*>(s_block);
        for (int i = tid; i < npos * 16; i += kSmallThreads) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    for (uint32_t r0 = (uint32_t)tid; r0 < n; r0 += kSmallThreads * kSmallRows) {
        uint32_t row[kSmallRows];
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) {
            const uint32_t r = r0 + (uint32_t)k * kSmallThreads;
            row[k] = r < n ? r : n - 1;
        }
        for (int c = 0; c < ncols; c++) {
            const DevCol& col = a.cols.c[c];
            const int p0 = (int)s_start[c], maxlen = (int)s_max[c];
            const uint64_t pa = (uint64_t)(uintptr_t)col.data;
            const uint8_t* base8 = (const uint8_t*)(uintptr_t)(pa & ~7ull);
            const uint32_t delta = (uint32_t)(pa & 7ull);
            uint64_t b[kSmallRows];
            uint32_t l[kSmallRows];
            small_spans(col, row, b, l);
            for (int j = 0; 8 * j < maxlen; j++) {
                uint64_t ch[kSmallRows];
#pragma unroll
                for (int k = 0; k < kSmallRows; k++) ch[k] = load_chunk_nobranch<uint64_t>(base8, delta, b[k], l[k], (uint32_t)j);
#pragma unroll
                for (int bb = 0; bb < 8; bb++) {
                    const int q = 8 * j + bb;
                    if (q >= maxlen) break;   // uniform
#pragma unroll
                    for (int k = 0; k < kSmallRows; k++)
                        if ((uint32_t)q < l[k]) s_block[(p0 + q) * 256 + (int)((ch[k] >> (8 * bb)) & 0xFF)] = 1;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) s_t[2] = wall_clock64();
    // masks for the host (ColStats::mask of position p, word i): kept in LDS, written out with the rest of the result
    for (int i = tid; i < npos * 8; i += kSmallThreads) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(s_block + 32 * i);
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t v = f[k];
            bits |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * k);
        }
        s_mask[i] = bits;
    }
    {   // codec_build (keycodec.hip): pad first when some value ends before q, then the bytes present, in byte order;
        // one wave per position: lane i ranks bytes 4 i .. 4 i + 3 behind the bytes of the lanes below
        for (int p = w; p < npos; p += kSmallWaves) {
            int c = 0;
            while (c + 1 < ncols && (uint32_t)p >= s_start[c + 1]) c++;
            const uint32_t q = (uint32_t)p - s_start[c];
            const uint32_t f4 = reinterpret_cast<const uint32_t*>(s_block + p * 256)[lane];   // flags of 4 byte values
            const uint32_t mine = (f4 & 1u) + ((f4 >> 8) & 1u) + ((f4 >> 16) & 1u) + ((f4 >> 24) & 1u);
            const uint32_t pad = q >= s_min[c] ? 1u : 0u;
            uint32_t rank = wave_inclusive_sum(mine) - mine + pad;
            uint16_t* lut = &s_lut[p * kLutStride];
            if (lane == 0) lut[0] = pad ? (uint16_t)0 : kLutInvalid;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool present = (f4 >> (8 * k)) & 1u;
                lut[1 + 4 * lane + k] = present ? (uint16_t)rank : kLutInvalid;
                rank += present ? 1u : 0u;
            }
            if (lane == kWave - 1) s_radix[p] = (uint16_t)rank;
        }
    }
    __syncthreads();
    if (tid == 0) {   // codec_split_words: one word of at most 2^63 states, weights from the last position up
        uint64_t prod = 1;
        bool fits = true;
        for (int p = npos - 1; p >= 0; p--) {
            s_mult[p] = prod;
            const uint64_t r = s_radix[p];
            const uint64_t hi = __umul64hi(prod, r), lo = prod * r;
            if (hi != 0 || lo > (1ull << 63)) { fits = false; break; }
            prod = lo;
        }
        if (!fits) s_status = kSmallNotSmall;
        s_bits = small_bits_needed(prod);
        s_key32 = prod <= (1ull << 32) ? 1u : 0u;
    }
    __syncthreads();
    if (s_status != kSmallBuilt) {
        if (tid == 0) { a.res->status = kSmallNotSmall; __threadfence_system(); }
        return;
    }

    if (tid == 0) s_t[3] = wall_clock64();
    // ---- 3: encode ----
    uint64_t* kin = a.ka;
    uint64_t* kout = a.kb;
    uint32_t* vin = a.va;
    uint32_t* vout = a.vb;
    for (uint32_t r0 = (uint32_t)tid; r0 < n; r0 += kSmallThreads * kSmallRows) {
        uint32_t row[kSmallRows];
        uint64_t code[kSmallRows];
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) {
            const uint32_t r = r0 + (uint32_t)k * kSmallThreads;
            row[k] = r < n ? r : n - 1;
            code[k] = 0;
        }
        for (int c = 0; c < ncols; c++) {
            const DevCol& col = a.cols.c[c];
            const int p0 = (int)s_start[c], maxlen = (int)s_max[c];
            const uint64_t pa = (uint64_t)(uintptr_t)col.data;
            const uint8_t* base8 = (const uint8_t*)(uintptr_t)(pa & ~7ull);
            const uint32_t delta = (uint32_t)(pa & 7ull);
            uint64_t b[kSmallRows];
            uint32_t l[kSmallRows];
            small_spans(col, row, b, l);
            for (int j = 0; 8 * j < maxlen; j++) {
                uint64_t ch[kSmallRows];
#pragma unroll
                for (int k = 0; k < kSmallRows; k++) ch[k] = load_chunk_nobranch<uint64_t>(base8, delta, b[k], l[k], (uint32_t)j);
#pragma unroll
                for (int bb = 0; bb < 8; bb++) {
                    const int q = 8 * j + bb;
                    if (q >= maxlen) break;   // uniform
                    const uint64_t m = s_mult[p0 + q];
                    const uint16_t* lut = &s_lut[(p0 + q) * kLutStride];
#pragma unroll
                    for (int k = 0; k < kSmallRows; k++) {
                        const int sym = (uint32_t)q < l[k] ? (int)((ch[k] >> (8 * bb)) & 0xFF) + 1 : 0;
                        code[k] += (uint64_t)lut[sym] * m;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kSmallRows; k++) {
            const uint32_t r = r0 + (uint32_t)k * kSmallThreads;
            if (r < n) kin[r] = code[k];
        }
    }
    __syncthreads();

    if (tid == 0) s_t[4] = wall_clock64();
    // ---- 4: stable LSD sort, 8-bit digits (balanced over the passes) ----
    // wave w owns rows [c_begin, c_end): at most kSmallItems * 64 of them, all held in registers during a pass
    const int bits = (int)s_bits;
    const int npass = (bits + 7) / 8;
    uint32_t(*s_hist)[256] = reinterpret_cast<uint32_t(*)[256]>(s_block);
    const uint32_t chunk_rows = ((n + kSmallWaves - 1) / kSmallWaves + kWave - 1) / kWave * kWave;   // rows per wave, multiple of 64
    const uint32_t c_begin = (uint32_t)w * chunk_rows < n ? (uint32_t)w * chunk_rows : n;
    const uint32_t c_end = c_begin + chunk_rows < n ? c_begin + chunk_rows : n;
    const uint64_t lt = lanemask_lt();
    int shift = 0;
    for (int p = 0; p < npass; p++) {
        const int left = bits - shift;
        const int nb = (left + (npass - p) - 1) / (npass - p);
        const uint32_t dmask = (1u << nb) - 1u;
        for (int i = tid; i < kSmallWaves * 256; i += kSmallThreads) (&s_hist[0][0])[i] = 0;
        uint64_t key[kSmallItems];
        uint32_t val[kSmallItems];
#pragma unroll
        for (int k = 0; k < kSmallItems; k++) {
            const uint32_t i = c_begin + (uint32_t)k * kWave + (uint32_t)lane;
            const bool valid = i < c_end;
            key[k] = valid ? kin[i] : 0ull;
            val[k] = valid ? (p == 0 ? i : vin[i]) : 0u;
        }
        __syncthreads();
        // count: the lanes holding the same digit are found ONCE per key; its rank among them and (for the first of them)
        // their number are kept for the scatter below
        uint32_t info[kSmallItems];   // rank among the peers | peers << 8 (leader only, else 0)
#pragma unroll
        for (int k = 0; k < kSmallItems; k++) {
            info[k] = 0;
            if (c_begin + (uint32_t)k * kWave >= c_end) continue;   // wave-uniform
            const bool valid = c_begin + (uint32_t)k * kWave + (uint32_t)lane < c_end;
            const uint32_t d = valid ? (uint32_t)(key[k] >> shift) & dmask : 0u;
            const uint64_t peers = small_peers_nb(d, valid, nb);
            const uint32_t below = (uint32_t)__popcll(peers & lt);
            const bool leader = valid && below == 0;
            const uint32_t cnt = (uint32_t)__popcll(peers);
            info[k] = below | (leader ? cnt << 8 : 0u);
            if (leader) s_hist[w][d] += cnt;
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // offsets: digit-major over the whole table, wave-minor inside a digit
        {
            uint32_t run = 0;
            if (tid < 256)
                for (int ww = 0; ww < kSmallWaves; ww++) {
                    const uint32_t c = s_hist[ww][tid];
                    s_hist[ww][tid] = run;
                    run += c;
                }
            uint32_t total;
            const uint32_t start = block_exclusive_sum<uint32_t, kSmallThreads>(run, s_scan, &total);
            if (tid < 256)
                for (int ww = 0; ww < kSmallWaves; ww++) s_hist[ww][tid] += start;
        }
        __syncthreads();
        // scatter
#pragma unroll
        for (int k = 0; k < kSmallItems; k++) {
            if (c_begin + (uint32_t)k * kWave >= c_end) continue;   // wave-uniform
            const bool valid = c_begin + (uint32_t)k * kWave + (uint32_t)lane < c_end;
            const uint32_t d = valid ? (uint32_t)(key[k] >> shift) & dmask : 0u;
            const uint32_t before = s_hist[w][d];
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const uint32_t pos = before + (info[k] & 0xFFu);
                kout[pos] = key[k];
                vout[pos] = val[k];
                if (info[k] >> 8) s_hist[w][d] = before + (info[k] >> 8);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        shift += nb;
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }

    if (tid == 0) s_t[5] = wall_clock64();
    // ---- 5: results ----
    // the wave's rows once more, all loads in flight together; a row's predecessor is the lane below (or the last lane of
    // the previous item, or the last row of the previous wave's chunk)
    {
        const bool key32 = s_key32 != 0;
        uint64_t key[kSmallItems];
        uint32_t val[kSmallItems];
        const uint64_t before_chunk = c_begin > 0 && c_begin < n ? kin[c_begin - 1] : 0ull;
#pragma unroll
        for (int k = 0; k < kSmallItems; k++) {
            const uint32_t i = c_begin + (uint32_t)k * kWave + (uint32_t)lane;
            const bool valid = i < c_end;
            key[k] = valid ? kin[i] : 0ull;
            val[k] = valid ? (npass == 0 ? i : vin[i]) : 0u;
        }
        uint32_t best = 0xFFFFFFFFu;
        uint64_t carry = before_chunk;   // the key in front of this item's lane 0
#pragma unroll
        for (int k = 0; k < kSmallItems; k++) {
            const uint32_t i = c_begin + (uint32_t)k * kWave + (uint32_t)lane;
            const bool valid = i < c_end;
            const uint64_t up = __shfl_up(key[k], 1, kWave);
            const uint64_t prev = lane == 0 ? carry : up;
            carry = __shfl(key[k], kWave - 1, kWave);
            if (valid) {
                if (key32) reinterpret_cast<uint32_t*>(a.sorted)[i] = (uint32_t)key[k];
                else reinterpret_cast<uint64_t*>(a.sorted)[i] = key[k];
                a.perm[i] = val[k];
                if (i > 0 && prev == key[k] && i < best) best = i;
            }
        }
        best = wave_min(best);
        if (lane == 0 && best != 0xFFFFFFFFu) atomicMin(&s_dup, best);
    }
    lds_atomics_barrier();
    if (w == 0) {   // one wave talks to the host: result block first, then (behind a system-scope fence) the status word
        SmallResult* res = a.res;
        for (int i = lane; i < npos * 8; i += kWave) res->mask[i >> 3][i & 7] = s_mask[i];
        if (lane < ncols) { res->minlen[lane] = s_min[lane]; res->maxlen[lane] = s_max[lane]; }
        if (lane == 0) {
            s_t[6] = wall_clock64();
            for (int i = 0; i < 7; i++) res->t[i] = s_t[i];
            res->first_dup = s_dup;
            res->bits = s_bits;
            res->key32 = s_key32;
            res->passes = (uint32_t)npass;
            uint64_t h = 0;
            for (int p = 0; p < npos; p++) h = small_codec_check(h, s_radix[p], s_mult[p]);
            res->codec_check = h;
        }
        __threadfence_system();
        if (lane == 0) res->status = kSmallBuilt;
    }
}

bool small_build_applies(const cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n) {
    if (n < 1 || n > (uint64_t)ctx->small_build_rows || n > (uint64_t)kSmallMaxRows || ncols < 1 || ncols > kMaxKeyCols) return false;
    for (int c = 0; c < ncols; c++)
        if (cols[c].segmented()) return false;
    return true;
}

Status small_build_launch(cph_ctx* ctx, const DevCol* cols, int32_t ncols, uint64_t n, SmallBufs* bufs, SmallResult* res) {
    CPH_TRY(bufs->ka.alloc(&ctx->pool, n * sizeof(uint64_t)));
    CPH_TRY(bufs->kb.alloc(&ctx->pool, n * sizeof(uint64_t)));
    CPH_TRY(bufs->va.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(bufs->vb.alloc(&ctx->pool, n * sizeof(uint32_t)));
    CPH_TRY(bufs->sorted.alloc(&ctx->pool, n * sizeof(uint64_t)));
    CPH_TRY(bufs->perm.alloc(&ctx->pool, n * sizeof(uint32_t)));
    res->status = kSmallPending;
    SmallArg a;
    for (int c = 0; c < ncols; c++) a.cols.c[c] = cols[c];
    a.ncols = ncols;
    a.n = (uint32_t)n;
    a.ka = bufs->ka.as<uint64_t>();
    a.kb = bufs->kb.as<uint64_t>();
    a.va = bufs->va.as<uint32_t>();
    a.vb = bufs->vb.as<uint32_t>();
    a.sorted = bufs->sorted.get();
    a.perm = bufs->perm.as<uint32_t>();
    a.res = res;
    ProfScope ps(ctx, "k_small_build", 0);
    hipLaunchKernelGGL(k_small_build, dim3(1), dim3(kSmallThreads), 0, ctx->stream, a);
    CPH_HIP_TRY(hipGetLastError());
    return {};
}

// After the stream has been synchronised: the index from the result block, or *not_small when the key needs the
// general path.  The codec is rebuilt on the host from the statistics the kernel gathered (the same inputs give the
// same LUT and weights: codec_build) and must agree with what the kernel sorted by.
Status small_build_finish(cph_ctx* ctx, cph_index* ix, int32_t ncols, SmallBufs* bufs, const SmallResult* res, bool* not_small) {
    *not_small = false;
    if (res->status == kSmallNotSmall) { *not_small = true; return {}; }
    if (res->status != kSmallBuilt) return {CPH_ERR_HIP, "internal: the small-table build kernel left no result"};
    if (ctx->codec_debug)   // 100 MHz ticks: lengths | flags | LUT | weights+encode | sort | results
        fprintf(stderr, "k_small_build n=%llu bits=%u passes=%u phases(us): len %.2f flags %.2f lut %.2f encode %.2f sort %.2f out %.2f\n",
                (unsigned long long)ix->nrows, res->bits, res->passes, (res->t[1] - res->t[0]) / 100.0, (res->t[2] - res->t[1]) / 100.0,
                (res->t[3] - res->t[2]) / 100.0, (res->t[4] - res->t[3]) / 100.0, (res->t[5] - res->t[4]) / 100.0, (res->t[6] - res->t[5]) / 100.0);
    std::vector<ColStats> stats((size_t)ncols);
    uint32_t p0 = 0;
    for (int c = 0; c < ncols; c++) {
        ColStats& s = stats[(size_t)c];
        memset(&s, 0, sizeof s);
        s.minlen = res->minlen[c];
        s.maxlen = res->maxlen[c];
        if (p0 + s.maxlen > (uint32_t)kSmallMaxPos) return {CPH_ERR_HIP, "internal: small-table statistics out of range"};
        memcpy(s.mask, res->mask[p0], sizeof(uint32_t) * 8 * s.maxlen);
        p0 += s.maxlen;
    }
    CPH_TRY(codec_build(stats, &ix->codec));
    uint64_t check = 0;
    for (int p = 0; p < ix->codec.npos && ix->codec.nwords == 1; p++) check = small_codec_check(check, ix->codec.radix[(size_t)p], ix->codec.mult[(size_t)p]);
    if (ix->codec.nwords != 1 || (ix->codec.key32 ? 1u : 0u) != res->key32 || (uint32_t)ix->codec.word_bits[0] != res->bits ||
        check != res->codec_check) {
        // cannot happen (same construction on both sides) — but if the two ever drift, the sorted codes and every later probe
        // would disagree silently: the general path is always right
        *not_small = true;
        ix->codec = CodecHost{};
        return {};
    }
    CPH_TRY(codec_upload(ctx, ix->codec, &ix->codec_dev));
    ix->sorted_codes = std::move(bufs->sorted);
    ix->perm = std::move(bufs->perm);
    ix->sort_passes = (int32_t)res->passes;
    ix->small_built = true;
    ix->first_dup = res->first_dup != 0xFFFFFFFFu ? (uint64_t)res->first_dup : UINT64_MAX;
    return {};
}

void warm_small_build() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_small_build));
    (void)hipGetLastError();
}

}  // namespace cph

// chain.hip — chained Join on the device:  stream.Join(a, ka).Join(b, kb)...  (README.md:56,
// csvplus.go:545-569 nested: the second Join's source IS the first Join's closure), for the
// case where every key comes from a column of the stream table (mergeRows, csvplus.go:571-583,
// lets the stream's value win on a name collision, so that is the value the next Join sees).
//
// Fast path: every index has distinct keys, a single key column and a one-word code with a
// pre-multiplied LUT.  Then a stream row joins at most once, and the output is written
// OPTIMISTICALLY DENSE:
//   k_chain_dense   one pass over the stream rows.  Per row and step: encode the key (one LDS
//                   load + add per byte position), look the code up in the step's direct-address
//                   table (or binary-search the sorted codes).  A row that matched in every step
//                   writes its tuple at slot == its row number; the per-wave match ballots and the
//                   per-tile match counts are recorded.  No workgroup talks to another one.
//                   Each thread keeps kChainRows rows in flight: all offset loads, then all key
//                   loads, then all table loads are issued back to back (the lookups are
//                   latency-bound otherwise).
//   k_sum_counts    total number of joined rows.
//   if total == rows (every stream row joined — the README chain, BASELINE config 4): done, the
//   dense arrays ARE the result in emission order (and result row m is stream row probe_base+m, so
//   no stream_row array is materialised at all: cph_chain.stream_row == NULL);
//   else            exclusive scan of the tile counts + k_chain_compact moves the matched tuples
//                   to their final slots (stream order is kept: slot = matches before the row).
// An earlier single-pass variant (decoupled look-back over an atomic tile ticket) was measured at
// 2.3-3.7 ms per 1e8 rows: one device-wide ticket counter sustains ~88 increments/us
// (MI355X_MICROARCH.md "dequeue"), which alone capped it.
//
// Algorithmic traffic per stream row: sum over steps of (key bytes + offset) in, one 8-byte table
// entry per step (random access), 4*steps bytes out per joined row (when compaction is needed: the
// same again read, plus 8 + 4*steps written per surviving row).
//
// General path (duplicate keys / multi-column keys / multi-word codes): probe, select, compose
// step by step with the generic kernels of probe.hip.
#include <type_traits>

#include "probe_device.hpp"

namespace cph {

constexpr int kChainThreads = 256;
constexpr int kChainWaves   = kChainThreads / kWave;
constexpr int kChainRows    = 8;                                   // rows in flight per lane
constexpr int kWaveTile     = kWave * kChainRows;                  // consecutive stream rows owned by one wave
constexpr int kChainTile    = kChainThreads * kChainRows;          // stream rows per workgroup tile
constexpr int kChainMasks   = kChainRows * kChainWaves;            // ballot words per tile (a plain bitmap: word r/64)
constexpr int kMaxChain     = 4;                                   // Joins of one FUSED pass (the kernels' argument blocks); CPH_MAX_CHAIN (8) is the limit of a call

struct ChainStepArg {
    DevCol col;                 // the stream's key column for this step
    const uint8_t* codec;       // codec block of the step's index (global memory)
    const uint32_t* rowtab;     // code -> build row (0xFFFFFFFF: absent), or nullptr
    const uint2* ranktab;       // positions mode: {present bits, keys before} per 32 codes (probe.hip: k_build_ranktab), or nullptr
    int32_t positions;          // the step reports the key's SORTED POSITION instead of the build row
    int32_t ranktab_lds;        // != 0: PAIRS of blocks of ranktab every workgroup copies into LDS (a small index: no L2 -> L1 line per row)
    const uint4* hash;          // no rowtab: hash table over the codes (hash_device.hpp, kHashK1 entries), or nullptr -> binary search
    uint32_t hash_sectors;
    uint32_t hash_slice_mask;
    int32_t src;                // cph_chain_step.source: 0 = col belongs to the stream; k / -k = to the build table of step k-1 (DEP kernels only)
    const uint32_t* src_perm;   // src > 0 in positions mode: the source index's perm (sorted position -> original row), else nullptr
    const void* codes;          // sorted codes (u32 if key32 else u64)
    const uint32_t* perm;
    uint64_t n_index;
    int32_t codec_bytes;        // 0: a lean step (k_chain_dense's LEAN mask): no codec block is copied into LDS
    int32_t key32;
};
struct ChainArgs {
    ChainStepArg step[kMaxChain];
    uint32_t* out_rows[kMaxChain];
    int32_t nt_streams;         // the stream's offsets / key bytes are loaded and the results stored non-temporally: they pass
                                // through once and must not push the lookup tables out of the L2
    int32_t reserved_;
};
// what only the lean kernels read (LEAN != 0), as the kernel's LAST argument
struct LeanArgs {
    int32_t identity[kMaxChain];   // duplicate-free index whose code space has exactly as many states as the index has rows:
                                   // every code occurs, the sorted position of a key IS its code — no lookup at all
    ArithPlan arith[kMaxChain];    // lean steps: the key code as a dot product of the key's bytes (codec_device.hpp)
};

// DBG: attribution switches for tools/microbench (results are wrong when set):
//      dbg & 1 = no table lookup, & 2 = no encode, & 4 = no output stores.  LONG: some step's
//      index has keys longer than 8 bytes (then bytes 8..15 are prefetched too).  WIDE: 64-bit value
//      offsets or 64-bit codes somewhere in the chain (otherwise both stay in one 32-bit register).
// A wave owns kWaveTile consecutive stream rows: row(k, lane) = wave base + 64 k + lane, so every memory
// instruction of a phase covers 64 neighbouring rows.  Each phase — value spans, key bytes, LUT walk, table
// lookups — is straight-line code over all kChainRows rows and all S steps: rows past the end are clamped to the
// last row and unmatched codes to entry 0 (results masked afterwards) instead of being branched around, so the
// kChainRows * S loads of a phase are in flight together.  No workgroup-level synchronisation in the tile loop.
// LEAN: bit s set = step s is a "lean" step, decided on the host (enqueue_dense): its stream column is fixed-width 8 at an
//      8-byte aligned address (a value is ONE aligned 8-byte load) and its code is a dot product of the key's bytes
//      (ArithPlan) — no codec block in LDS, no LUT walk, no length compare.  A compile-time property: the other steps'
//      code is not even instantiated for it (fewer live SGPRs / VGPRs than a uniform branch leaves behind).  Only
//      instantiated for !LONG && !WIDE && !DBG.
// R: rows per lane and phase (8, or 4 for the register-heavy instantiations — long keys AND 64-bit codes in a chain of two or
//      more steps need 256+ VGPRs at 8 rows: one wave per SIMD, and a hash probe's sector loads want many waves in flight).
//      The wave still owns kWaveTile rows per tile: it walks them in kChainRows / R parts, so the tile geometry, the match
//      bitmap and the per-(tile, wave) counts are the same for every R.
// DEP: some step reads its key from the BUILD TABLE of an earlier step (cph_chain_step.source != 0: people.Join(orders, "id")
//      .Join(products), csvplus_test.go:280-285 — prod_id is a column of the orders row the first Join matched, which mergeRows
//      copied into the row the second Join sees, csvplus.go:571-583).  The phases then run step by step instead of all steps
//      per phase: step s gathers its value from row brow[source][k] of its column (through the source index's perm when the
//      chain reports positions and the column is in the table's original order), so a step starts when its source step's
//      lookups are back.  Instantiated for LONG && WIDE && !LEAN, R = 4 only (the general register types).
template <int S, bool LONG, bool WIDE, bool DBG, uint32_t LEAN = 0, int R = kChainRows, bool DEP = false>
__global__ __launch_bounds__(kChainThreads) void k_chain_dense(ChainArgs a, uint64_t nprobe, uint64_t probe_base,
                                                              uint64_t ntiles, uint64_t* __restrict__ masks,
                                                              uint32_t* __restrict__ wave_counts, int dbg_flags, LeanArgs la) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // dynamic LDS layout: [codec 0][codec 1]...   (no static LDS: keeps 16-B alignment)
    using B = std::conditional_t<WIDE, uint64_t, uint32_t>;    // byte offset of a value from its (uniform) base
    using CW = std::conditional_t<WIDE, uint64_t, uint32_t>;   // key code
    const int dbg = DBG ? dbg_flags : 0;
    CodecView cv[S];
    {
        uint8_t* p = smem;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if ((LEAN >> s) & 1u) continue;   // codec_bytes == 0: nothing of it is needed
            cv[s] = codec_load_to_lds(a.step[s].codec, p);   // syncs inside
            p += a.step[s].codec_bytes;
        }
    }
    // rank tables small enough to live in LDS next to the codecs (the 1e5-row products index: 2517 blocks), 12 bytes
    // per block there: {bits lo, bits hi, keys before}
    const CPH_LDS uint32_t* rank_lds[S];
    {
        uint8_t* p = smem;
#pragma unroll
        for (int s = 0; s < S; s++) p += a.step[s].codec_bytes;
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; s++) {
            rank_lds[s] = nullptr;
            const int nblk = a.step[s].ranktab_lds;
            if (LEAN != 0 && nblk) {   // the blocks as they are: 8 bytes per 32 codes, one ds_read_b64 and 32-bit arithmetic per lookup
                uint2* dst = reinterpret_cast<uint2*>(p);
                for (int i = threadIdx.x; i < 2 * nblk; i += kChainThreads) dst[i] = a.step[s].ranktab[i];
                rank_lds[s] = (const CPH_LDS uint32_t*)p;
                p += (size_t)nblk * 16;
                any = true;
            } else if (nblk) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(p);
                for (int i = threadIdx.x; i < nblk; i += kChainThreads) {   // two 32-code blocks -> {bits, bits, keys before}
                    const uint2 b0 = a.step[s].ranktab[2 * i], b1 = a.step[s].ranktab[2 * i + 1];
                    dst[3 * i] = b0.x;
                    dst[3 * i + 1] = b1.x;
                    dst[3 * i + 2] = b0.x ? b0.y : b1.y;   // an empty block carries no count: the pair's keys all sit in the other one
                }
                rank_lds[s] = (const CPH_LDS uint32_t*)p;
                p += (size_t)((nblk * 12 + 15) & ~15);
                any = true;
            }
        }
        if (any) __syncthreads();
    }
    const int lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(wave_id());

#pragma unroll 1
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t wave_matches = 0;
#pragma unroll 1
      for (int part = 0; part < kChainRows / R; part++) {
        const uint64_t wbase = tile * kChainTile + (uint64_t)wave * kWaveTile + (uint64_t)part * (R * kWave);   // wave-uniform
        const WaveRows<R> wr = wave_rows<R>(wbase, nprobe);
        uint32_t okm = wr.okm;                  // bit k: row k of this lane is (still) joined
        WaveSpans<R, B> sp[S];
        uint64_t c0[S][R], c1[LONG ? S : 1][R];
        CW code[S][R];
        uint32_t brow[S][R];
        // !DEP: ONE level, every phase covers all steps.  DEP: level l is step l alone (its source step's rows are known by then).
#pragma unroll
      for (int lvl = 0; lvl < (DEP ? S : 1); lvl++) {
        // ---- A: value spans ----------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (DEP && s != lvl) continue;
            if ((LEAN >> s) & 1u) continue;   // a lean step needs no spans: value k is the 8-byte word rbase + rel[k]
            if constexpr (DEP) {
                const int32_t src = a.step[s].src;
                if (src != 0) {   // uniform: the key sits in the row an earlier step matched
                    const int t = (src < 0 ? -src : src) - 1;
                    uint32_t grow[R];
#pragma unroll
                    for (int k = 0; k < R; k++) grow[k] = 0u;
#pragma unroll
                    for (int u = 0; u < S; u++)
                        if (u < s && u == t) {
#pragma unroll
                            for (int k = 0; k < R; k++) grow[k] = (okm >> k) & 1u ? brow[u][k] : 0u;   // row 0 exists (no index of the chain is empty)
                        }
                    if (a.step[s].src_perm) {
#pragma unroll
                        for (int k = 0; k < R; k++) grow[k] = a.step[s].src_perm[grow[k]];
                    }
                    gather_spans<R, B>(a.step[s].col, grow, &sp[s]);
                    continue;
                }
            }
            if (a.nt_streams) wave_spans<R, B, false, true>(a.step[s].col, wr, &sp[s]);   // uniform branch
            else wave_spans<R, B>(a.step[s].col, wr, &sp[s]);
        }
        // ---- B: first 8 (16) key bytes ---------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (DEP && s != lvl) continue;
            if ((LEAN >> s) & 1u) {
                typedef const __attribute__((address_space(1))) uint64_t* global_u64_ptr;
                const global_u64_ptr w = (global_u64_ptr)a.step[s].col.data + wr.rbase;
                if (a.nt_streams) {   // uniform, outside the row loop
#pragma unroll
                    for (int k = 0; k < R; k++) c0[s][k] = __builtin_nontemporal_load(&w[wr.rel[k]]);
                } else {
#pragma unroll
                    for (int k = 0; k < R; k++) c0[s][k] = w[wr.rel[k]];
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < R; k++) {
                c0[s][k] = a.nt_streams ? sp[s].chunk_nt(k, 0) : sp[s].chunk(k, 0);
                if constexpr (LONG) c1[s][k] = a.nt_streams ? sp[s].chunk_nt(k, 1) : sp[s].chunk(k, 1);
            }
        }
        // ---- C: codes -----------------------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (DEP && s != lvl) continue;
            if (DBG && (dbg & 2)) {
#pragma unroll
                for (int k = 0; k < R; k++) code[s][k] = (CW)((c0[s][k] ^ (LONG ? c1[LONG ? s : 0][k] : 0ull)) & 1023);
            } else if ((LEAN >> s) & 1u) {   // the code is a dot product of the key's bytes
                encode_rows_arith<R, CW>(la.arith[s], c0[s], code[s], &okm);
            } else if (!WIDE || cv[s].hdr->lutw_bits == 32) {
                encode_rows<R, uint32_t, B, CW, LONG>(cv[s], sp[s], c0[s], c1[LONG ? s : 0], code[s], &okm,
                                                               a.step[s].col.fixed_width != 0 && (int)a.step[s].col.fixed_width == cv[s].hdr->col_maxlen[0]);
            } else {
                encode_rows<R, uint64_t, B, CW, LONG>(cv[s], sp[s], c0[s], c1[LONG ? s : 0], code[s], &okm,
                                                               a.step[s].col.fixed_width != 0 && (int)a.step[s].col.fixed_width == cv[s].hdr->col_maxlen[0]);
            }
        }
        // ---- D: lookups ---------------------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (DEP && s != lvl) continue;
            const ChainStepArg& st = a.step[s];
            if constexpr (LEAN != 0) {
                // a lean kernel reports positions and every step is answered by one of three lookups (enqueue_dense checks)
                if (la.identity[s]) {   // every code below n_index occurs and is its own sorted position: no lookup at all
#pragma unroll
                    for (int k = 0; k < R; k++)
                        brow[s][k] = ((okm >> k) & 1u) && (uint32_t)code[s][k] < (uint32_t)st.n_index ? (uint32_t)code[s][k] : kTableAbsent;
                } else if (rank_lds[s]) {   // 8-byte rank blocks in LDS
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const uint32_t cidx = (okm >> k) & 1u ? (uint32_t)code[s][k] : 0u;
                        const CPH_LDS uint32_t* e = rank_lds[s] + 2u * (cidx >> 5);
                        const uint32_t bits = e[0], before = e[1], bit = cidx & 31u;
                        brow[s][k] = (bits >> bit) & 1u ? before + (uint32_t)__popc(bits & ((1u << bit) - 1u)) : kTableAbsent;
                    }
                } else {             // rank blocks in global memory (L2-resident for a 1e7-row index)
                    uint2 blk[R];
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const uint32_t cidx = (okm >> k) & 1u ? (uint32_t)code[s][k] : 0u;   // block 0 always exists
                        blk[k] = st.ranktab[cidx >> 5];
                    }
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const uint32_t bit = (uint32_t)code[s][k] & 31u;
                        brow[s][k] = (blk[k].x >> bit) & 1u ? blk[k].y + (uint32_t)__popc(blk[k].x & ((1u << bit) - 1u)) : kTableAbsent;
                    }
                }
            } else if (st.ranktab) {
                // the key's rank among the index keys = its sorted position: one 8-byte block per row from a table
                // 1/16 the size of rowtab (L2-resident for the 1e7-row customers index: no Infinity-Fabric sector per
                // row), or 12 bytes per 64 codes from LDS
                if (rank_lds[s]) {   // uniform branch
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const CW cidx = (okm >> k) & 1u ? code[s][k] : (CW)0;
                        const CPH_LDS uint32_t* e = rank_lds[s] + 3u * (uint32_t)((uint64_t)cidx >> 6);
                        const uint32_t bit = (uint32_t)cidx & 63u;
                        const uint64_t bits = (uint64_t)e[0] | ((uint64_t)e[1] << 32);
                        const bool present = (bits >> bit) & 1ull;
                        brow[s][k] = present ? e[2] + (uint32_t)__popcll(bits & ((1ull << bit) - 1ull)) : kTableAbsent;
                    }
                } else {
                    uint2 blk[R];
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const CW cidx = (okm >> k) & 1u ? code[s][k] : (CW)0;   // block 0 always exists
                        blk[k] = (DBG && (dbg & 1)) ? make_uint2(~0u, 0u) : st.ranktab[(uint64_t)cidx >> 5];
                    }
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const uint32_t bit = (uint32_t)code[s][k] & 31u;
                        const bool present = (blk[k].x >> bit) & 1u;
                        brow[s][k] = present ? blk[k].y + (uint32_t)__popc(blk[k].x & ((1u << bit) - 1u)) : kTableAbsent;
                    }
                }
            } else if (st.rowtab) {
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const CW cidx = (okm >> k) & 1u ? code[s][k] : (CW)0;   // entry 0 always exists
                    brow[s][k] = (DBG && (dbg & 1)) ? 0u : st.rowtab[cidx];
                }
            } else if (st.hash) {
                // sparse code space: the home sectors of 4 rows at a time are loaded together (16 x 16 bytes in flight
                // per lane); entry.aux is the build row (duplicate-free index)
                const HashView hv{st.hash, st.hash_sectors, st.hash_slice_mask};
#pragma unroll
                for (int g = 0; g < R; g += 4) {
                    HashSector sc[4];
                    uint32_t home[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        home[k] = hash_home(hash_one((uint64_t)code[s][g + k]), hv.nsectors);
                        sc[k] = hash_load_sector(hv, (okm >> (g + k)) & 1u ? home[k] : 0u);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t l, a;
                        bool more;
                        bool hit = hash_match16(sc[k], (uint64_t)code[s][g + k], &l, &a, &more);
                        const bool live = (okm >> (g + k)) & 1u;
                        if (live && more) hit = hash_continue16(hv, home[k], (uint64_t)code[s][g + k], &l, &a);   // rare
                        brow[s][g + k] = (hit && live && !(DBG && (dbg & 1))) ? (st.positions ? l : a) : kTableAbsent;
                        if (DBG && (dbg & 1)) brow[s][g + k] = 0u;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; k++) {
                    brow[s][k] = kTableAbsent;
                    if (!((okm >> k) & 1u)) continue;
                    uint64_t lo;
                    bool hit;
                    if (st.key32) {
                        const uint32_t* cd = reinterpret_cast<const uint32_t*>(st.codes);
                        lo = lower_bound_dev<uint32_t>(cd, 0, st.n_index, (uint32_t)code[s][k]);
                        hit = lo < st.n_index && cd[lo] == (uint32_t)code[s][k];
                    } else {
                        const uint64_t* cd = reinterpret_cast<const uint64_t*>(st.codes);
                        lo = lower_bound_dev<uint64_t>(cd, 0, st.n_index, (uint64_t)code[s][k]);
                        hit = lo < st.n_index && cd[lo] == (uint64_t)code[s][k];
                    }
                    if (hit) brow[s][k] = st.positions ? (uint32_t)lo : st.perm[lo];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (DEP && s != lvl) continue;
#pragma unroll
            for (int k = 0; k < R; k++)
                if (brow[s][k] == kTableAbsent) okm &= ~(1u << k);
        }
      }
        // ---- dense output + match bookkeeping ----------------------------------------------------------
        const uint64_t mword = (tile * kChainWaves + wave) * kChainRows + (uint64_t)part * R;   // == wbase / 64
#pragma unroll
        for (int k = 0; k < R; k++) {
            const bool ok = (okm >> k) & 1u;
            const uint64_t bal = __ballot(ok);
            wave_matches += (uint32_t)__popcll(bal);
            if (lane == 0) masks[mword + k] = bal;
            // the stream row of slot r is probe_base + r by construction: it is never stored here
            if (ok && !(DBG && (dbg & 4))) {
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (a.nt_streams) __builtin_nontemporal_store(brow[s][k], a.out_rows[s] + wr.rbase + wr.rel[k]);
                    else (a.out_rows[s] + wr.rbase)[wr.rel[k]] = brow[s][k];
                }
            }
        }
      }
        // per-(tile, wave) match count
        if (lane == 0) wave_counts[tile * kChainWaves + wave] = wave_matches;
    }
}

// The same dense pass over key CODES the host formed (host_encode.hip: cph_host_encoder_run; a stream in host memory then
// ships 4 bytes per row and step instead of its key strings): no decode at all, one 4-byte load and one lookup per row and
// step.  Outputs and bookkeeping exactly as k_chain_dense (slot == row, match ballots, per-(tile, wave) counts).
struct CodeStepArg {
    const uint32_t* codes;      // the chunk's codes for this step (CPH_CODE_ABSENT: the key cannot be in the index)
    const uint32_t* rowtab;     // row ids: code -> build row
    const uint2* ranktab;       // positions: presence bits + keys before per 32 codes
    uint64_t n_index, table_entries;
    int32_t identity;           // positions, states == rows: the position is the code
    int32_t reserved_;
};
struct CodeArgs {
    CodeStepArg step[kMaxChain];
    uint32_t* out_rows[kMaxChain];
};
template <int S>
__global__ __launch_bounds__(kChainThreads) void k_chain_codes(CodeArgs a, uint64_t nprobe, uint64_t ntiles, uint64_t* __restrict__ masks,
                                                              uint32_t* __restrict__ wave_counts) {
    const int lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(wave_id());
#pragma unroll 1
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t wbase = tile * kChainTile + (uint64_t)wave * kWaveTile;
        const WaveRows<kChainRows> wr = wave_rows<kChainRows>(wbase, nprobe);
        uint32_t okm = wr.okm;
        uint32_t code[S][kChainRows], brow[S][kChainRows];
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int k = 0; k < kChainRows; k++) code[s][k] = (a.step[s].codes + wr.rbase)[wr.rel[k]];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const CodeStepArg& st = a.step[s];
            if (st.identity) {
#pragma unroll
                for (int k = 0; k < kChainRows; k++) brow[s][k] = (uint64_t)code[s][k] < st.n_index ? code[s][k] : kTableAbsent;
            } else if (st.ranktab) {
                uint2 blk[kChainRows];
#pragma unroll
                for (int k = 0; k < kChainRows; k++) {
                    const bool in = (uint64_t)code[s][k] < st.table_entries;
                    blk[k] = st.ranktab[in ? code[s][k] >> 5 : 0u];
                    if (!in) blk[k].x = 0u;
                }
#pragma unroll
                for (int k = 0; k < kChainRows; k++) {
                    const uint32_t bit = code[s][k] & 31u;
                    brow[s][k] = (blk[k].x >> bit) & 1u ? blk[k].y + (uint32_t)__popc(blk[k].x & ((1u << bit) - 1u)) : kTableAbsent;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kChainRows; k++) {
                    const bool in = (uint64_t)code[s][k] < st.table_entries;
                    const uint32_t r = st.rowtab[in ? code[s][k] : 0u];
                    brow[s][k] = in ? r : kTableAbsent;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int k = 0; k < kChainRows; k++)
                if (brow[s][k] == kTableAbsent) okm &= ~(1u << k);
        uint32_t wave_matches = 0;
        const uint64_t mword = (tile * kChainWaves + wave) * kChainRows;
#pragma unroll
        for (int k = 0; k < kChainRows; k++) {
            const bool ok = (okm >> k) & 1u;
            const uint64_t bal = __ballot(ok);
            wave_matches += (uint32_t)__popcll(bal);
            if (lane == 0) masks[mword + k] = bal;
            if (ok) {
#pragma unroll
                for (int s = 0; s < S; s++) (a.out_rows[s] + wr.rbase)[wr.rel[k]] = brow[s][k];
            }
        }
        if (lane == 0) wave_counts[tile * kChainWaves + wave] = wave_matches;
    }
}

// total += sum of the per-(tile, wave) counts (grid-stride; one atomic per workgroup)
__global__ __launch_bounds__(256) void k_sum_counts(const uint32_t* __restrict__ counts, uint64_t n,
                                                   unsigned long long* __restrict__ total) {
    __shared__ uint64_t s_w[256 / kWave];
    uint64_t t = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) t += counts[i];
    t = wave_sum(t);
    if (lane_id() == 0) s_w[wave_id()] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t r = 0;
        for (int w = 0; w < 256 / kWave; w++) r += s_w[w];
        if (r) atomicAdd(total, (unsigned long long)r);
    }
}

// The same, reporting straight to the host: the sum gathers in a SELF-CLEANING device accumulator (cph_ctx::SelfClean::sum:
// {u64 total, u32 ticket}, zero at rest) and the LAST workgroup stores it into *host_out (pinned host memory: cph::host_word)
// and zeroes the accumulator again — no memset in front of this launch, no device-to-host copy behind it.
__global__ __launch_bounds__(256) void k_sum_counts_report(const uint32_t* __restrict__ counts, uint64_t n, unsigned long long* __restrict__ acc,
                                                          uint32_t* __restrict__ ticket, unsigned long long* __restrict__ host_out) {
    __shared__ uint64_t s_w[256 / kWave];
    uint64_t t = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) t += counts[i];
    t = wave_sum(t);
    if (lane_id() == 0) s_w[wave_id()] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t r = 0;
        for (int w = 0; w < 256 / kWave; w++) r += s_w[w];
        if (r) atomicAdd(acc, (unsigned long long)r);
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1u) {   // every other workgroup's add happened before its ticket
            const unsigned long long total = atomicExch(acc, 0ull);
            atomicExch(ticket, 0u);
            *host_out = total;
            __threadfence_system();
        }
    }
}

// Moves the matched tuples of a wave's rows from slot == row to their final slots (wave_base = the exclusive scan
// of the (tile, wave) counts).  Wave-local: no LDS, no barrier.
template <int S>
__global__ __launch_bounds__(kChainThreads) void k_chain_compact(ChainArgs dense_rows, uint64_t probe_base,
                                                                const uint64_t* __restrict__ masks,
                                                                const uint32_t* __restrict__ wave_base, uint64_t ntiles,
                                                                uint64_t* __restrict__ out_stream, ChainArgs out_rows) {
    const int lane = lane_id(), wave = wave_id();
    const uint64_t lt = lanemask_lt();
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t wt = tile * kChainWaves + wave;
        uint64_t pos = wave_base[wt];
#pragma unroll
        for (int k = 0; k < kChainRows; k++) {
            const uint64_t m = masks[wt * kChainRows + k];
            if ((m >> lane) & 1ull) {
                const uint64_t row = wt * kWaveTile + (uint64_t)k * kWave + lane;
                const uint64_t p = pos + (uint64_t)__popcll(m & lt);
                out_stream[p] = probe_base + row;
#pragma unroll
                for (int s = 0; s < S; s++) out_rows.out_rows[s][p] = dense_rows.out_rows[s][row];
            }
            pos += (uint64_t)__popcll(m);
        }
    }
}

// ---- compose helpers for the general path ---------------------------------------------------------------
__global__ void k_compose_u64(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint64_t* __restrict__ dst,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}
__global__ void k_compose_u32(const uint32_t* __restrict__ src, const uint64_t* __restrict__ idx, uint32_t* __restrict__ dst,
                              uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

// dst[i] = perm[pos[i]]: sorted positions -> original rows (a later step reads its key from an earlier build table's columns)
__global__ void k_perm_rows(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ pos, uint32_t* __restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = perm[pos[i]];
}

// dst[i] = tab[idx[i]] (kTableAbsent stays kTableAbsent): a pre-joined table from the build table's row order into its sorted order
__global__ void k_gather_rows(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* __restrict__ dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = tab[idx[i]];
}

// Steps whose key is a column of an EARLIER step's build table (cph_chain_step.source != 0), answered from PRE-JOINED tables:
// tab[j][h] = what step dst[j] finds for the row with handle h (sorted position / original row) of its source step's table —
// the build sides were joined with each other first (run_prejoined), so per stream row such a step is ONE 4-byte gather instead
// of perm + offsets + key bytes + encode + lookup.  Runs behind k_chain_dense over the stream-keyed steps, in its geometry
// (a wave owns kWaveTile consecutive rows; masks = a plain bitmap; one count per (tile, wave)): clears the bit of a row whose
// step finds nothing and rewrites the counts.
struct PrejoinArgs {
    int32_t n;                       // dependent steps, in chain order
    int32_t src_dep[kMaxChain - 1];  // >= 0: the source is dependent step src_dep[j] of THIS list (its value is in registers); < 0: read src_rows[j]
    const uint32_t* src_rows[kMaxChain - 1];
    const uint32_t* tab[kMaxChain - 1];
    uint32_t* out_rows[kMaxChain - 1];
};
__global__ __launch_bounds__(kChainThreads) void k_chain_prejoined(PrejoinArgs a, uint64_t nprobe, uint64_t ntiles, uint64_t* __restrict__ masks,
                                                                  uint32_t* __restrict__ wave_counts) {
    const int lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(wave_id());
#pragma unroll 1
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t wbase = tile * kChainTile + (uint64_t)wave * kWaveTile;
        const uint64_t mword = (tile * kChainWaves + wave) * kChainRows;
        uint32_t okm = 0;
#pragma unroll
        for (int k = 0; k < kChainRows; k++) okm |= (uint32_t)((masks[mword + k] >> lane) & 1ull) << k;
        uint32_t val[kMaxChain - 1][kChainRows];
#pragma unroll
        for (int j = 0; j < kMaxChain - 1; j++) {
            if (j >= a.n) break;   // uniform
            uint32_t h[kChainRows];
            const int sd = a.src_dep[j];
            if (sd < 0) {
#pragma unroll
                for (int k = 0; k < kChainRows; k++) {
                    const uint64_t r = wbase + (uint64_t)k * kWave + lane;
                    h[k] = (okm >> k) & 1u ? a.src_rows[j][r] : 0u;   // (row 0 of a table exists: no index of the chain is empty)
                }
            } else {
#pragma unroll
                for (int k = 0; k < kChainRows; k++) h[k] = 0u;
#pragma unroll
                for (int u = 0; u < kMaxChain - 1; u++)
                    if (u < j && u == sd) {
#pragma unroll
                        for (int k = 0; k < kChainRows; k++) h[k] = (okm >> k) & 1u ? val[u][k] : 0u;
                    }
            }
#pragma unroll
            for (int k = 0; k < kChainRows; k++) val[j][k] = a.tab[j][h[k]];
#pragma unroll
            for (int k = 0; k < kChainRows; k++)
                if (val[j][k] == kTableAbsent) okm &= ~(1u << k);
        }
        uint32_t wave_matches = 0;
#pragma unroll
        for (int k = 0; k < kChainRows; k++) {
            const bool ok = (okm >> k) & 1u;
            const uint64_t bal = __ballot(ok);
            wave_matches += (uint32_t)__popcll(bal);
            if (lane == 0) masks[mword + k] = bal;
            if (ok) {
                const uint64_t r = wbase + (uint64_t)k * kWave + lane;
#pragma unroll
                for (int j = 0; j < kMaxChain - 1; j++)
                    if (j < a.n) __builtin_nontemporal_store(val[j][k], a.out_rows[j] + r);
            }
        }
        if (lane == 0) wave_counts[tile * kChainWaves + wave] = wave_matches;
    }
}


// ---- general path: a unique-index step keyed by an EARLIER BUILD TABLE, answered from a pre-joined table -----------------------
// (round 6) people.Join(IndexOn(orders.cust_id), "id").Join(products, prod_id of the ORDERS row) — csvplus_test.go:280-285 with a
// NON-unique first index (csvplus.go:559 emits every equal index row): the tuples in front of the second Join are pairs, the fast
// path does not apply, and per tuple the generic probe gathered perm + offsets + key bytes of the orders row (three random sectors,
// 4.5-5.2 ms per 1e8 tuples).  The table that step reads is joined with the products index ONCE instead (one dense, streaming pass over
// its nt rows: tab[row] = what the step finds), then a tuple costs one 4-byte gather; nothing is compacted when every tuple survives.
constexpr int kPjThreads = 256, kPjItems = 8, kPjTile = kPjThreads * kPjItems;
// perm != nullptr: the tuples hold sorted positions and the table is in its original row order (row = perm[position]; the positions of one
// stream row's matches are consecutive, so these reads are nearly sequential)
__global__ __launch_bounds__(kPjThreads) void k_prejoin_tuples(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ h,
                                                              const uint32_t* __restrict__ perm, uint64_t n,
                                                              uint32_t* __restrict__ out_v, uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t s_w[kPjThreads / kWave];
    const uint64_t base = (uint64_t)blockIdx.x * kPjTile;
    uint32_t hv[kPjItems], v[kPjItems];
#pragma unroll
    for (int k = 0; k < kPjItems; k++) {
        const uint64_t i = base + (uint64_t)k * kPjThreads + threadIdx.x;
        hv[k] = i < n ? __builtin_nontemporal_load(h + i) : 0u;   // (row 0 of the table exists)
    }
    if (perm) {   // uniform
#pragma unroll
        for (int k = 0; k < kPjItems; k++) hv[k] = perm[hv[k]];
    }
#pragma unroll
    for (int k = 0; k < kPjItems; k++) v[k] = tab[hv[k]];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kPjItems; k++) {
        const uint64_t i = base + (uint64_t)k * kPjThreads + threadIdx.x;
        if (i < n) {
            __builtin_nontemporal_store(v[k], out_v + i);
            cnt += v[k] != kTableAbsent ? 1u : 0u;
        }
    }
    cnt = wave_sum(cnt);
    if (lane_id() == 0) s_w[wave_id()] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kPjThreads / kWave; w++) t += s_w[w];
        tile_counts[blockIdx.x] = t;
    }
}
// the surviving tuples (v != absent) move to their final slots, in order: thread t owns kPjItems CONSECUTIVE tuples of the tile
struct PjCompactArgs {
    int32_t n32;
    const uint32_t* src32[CPH_MAX_CHAIN];
    uint32_t* dst32[CPH_MAX_CHAIN];
    const uint64_t* src64;
    uint64_t* dst64;
};
__global__ __launch_bounds__(kPjThreads) void k_prejoin_compact(PjCompactArgs a, const uint32_t* __restrict__ v, uint64_t n,
                                                               const uint32_t* __restrict__ tile_base) {
    __shared__ uint32_t s_tmp[kPjThreads / kWave + 1];
    const uint64_t i0 = (uint64_t)blockIdx.x * kPjTile + (uint64_t)threadIdx.x * kPjItems;
    uint32_t keep = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kPjItems; k++)
        if (i0 + k < n && v[i0 + k] != kTableAbsent) {
            keep |= 1u << k;
            mine++;
        }
    uint32_t total;
    uint32_t run = block_exclusive_sum<uint32_t, kPjThreads>(mine, s_tmp, &total);   // syncs inside
    uint64_t o = (uint64_t)tile_base[blockIdx.x] + run;
#pragma unroll
    for (int k = 0; k < kPjItems; k++) {
        if (!((keep >> k) & 1u)) continue;
        a.dst64[o] = a.src64[i0 + k];
        for (int j = 0; j < a.n32; j++) a.dst32[j][o] = a.src32[j][i0 + k];
        o++;
    }
}

bool chain_fast_path_ok(const ChainStep* steps, int nsteps) {
    if (nsteps > kMaxChain) return false;   // longer chains: the general path below, on the device, in the same call
    for (int s = 0; s < nsteps; s++) {
        const cph_index* ix = steps[s].index;
        if (steps[s].ncols != 1 || ix->nkeycols != 1) return false;
        if (ix->first_dup != UINT64_MAX) return false;
        if (ix->codec.nwords != 1 || ix->codec.npos == 0 || !ix->windows.empty()) return false;
        if (codec_premultiplied_bits(ix->codec) == 0) return false;
    }
    return true;
}

// Enqueues the dense pass + the match total on ctx->stream; nothing here waits for the GPU.
//   d_rows[s]  u32[nprobe]             build row of step s at slot == stream row (valid where the bit is set)
//   d_masks    u64[chain_dense_mask_words(nprobe)]  bit r%64 of word r/64 == "stream row r joined" (a plain bitmap)
//   d_counts   u32[chain_dense_count_words(nprobe)] matches per (tile, wave), tile-major
//   d_total    u64                     number of joined rows
template <int S>
static Status enqueue_dense(cph_ctx* ctx, const ChainStep* steps, uint64_t nprobe, uint64_t probe_base,
                            uint32_t* const* d_rows, uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total,
                            ChainArgs* args_out, unsigned* grid_out, bool positions, uint64_t* host_total = nullptr,
                            const char* prof_name = "k_chain_dense") {
    const uint64_t ntiles = (nprobe + kChainTile - 1) / kChainTile;
    ChainArgs args{};
    LeanArgs largs{};
    size_t lds = 0, rank_lds_bytes = 0;
    const int dbg = ctx->chain_debug;
    bool long_keys = false, wide = false, dep = false;
    for (int s = 0; s < S; s++) dep |= steps[s].source != 0;
    if (dep) long_keys = wide = true;   // the DEP kernels exist in the general register types only (and are never lean)
    for (int s = 0; s < S; s++) {
        const cph_index* ix = steps[s].index;
        const DevCol& c = steps[s].cols[0];
        long_keys |= ix->codec.col_maxlen[0] > 8;
        // 32-bit registers for value offsets and codes need: 32-bit offsets (addresses are formed in 64 bits at the
        // load), a 32-bit pre-multiplied LUT, a code that fits 32 bits
        wide |= codec_premultiplied_bits(ix->codec) != 32 || !ix->codec.key32;
        wide |= !col_is_narrow(c);
    }
    // lean steps (k_chain_dense's LEAN mask): a fixed-width-8 stream column at an 8-byte aligned address joined with an
    // index whose 8-byte keys code arithmetically (codec_arith_plan: contiguous alphabets, no pad).  Every subset of the
    // steps of chains of one or two Joins is instantiated; longer chains are lean in all steps or in none.
    // A lean kernel reports POSITIONS and answers every step from the code itself (identity) or from a rank table.
    uint32_t lean = 0;
    bool ident[kMaxChain] = {false, false, false, false};
    if (positions)
        for (int s = 0; s < S; s++) {
            const cph_index* ix = steps[s].index;
            ident[s] = ctx->chain_identity && ix->table_entries != 0 && ix->table_entries == ix->nrows &&
                       ix->first_dup == UINT64_MAX && ix->windows.empty();
        }
    if (positions && ctx->chain_arith && !long_keys && !wide && !dbg) {
        bool all_ranked = true;
        for (int s = 0; s < S; s++) {
            const cph_index* ix = steps[s].index;
            const DevCol& c = steps[s].cols[0];
            ArithPlan ap;
            codec_arith_plan(ix->codec, &ap);
            if (ap.enabled && ap.keylen == 8 && c.fixed_width == 8 && ((uintptr_t)c.data & 7u) == 0) lean |= 1u << s;
            if (!ident[s]) {
                CPH_TRY(index_ensure_ranktab(ctx, ix));
                all_ranked &= (bool)ix->ranktab;
            }
        }
        if (!all_ranked) lean = 0;
    }
    if (S > 2 && lean != (1u << S) - 1u) lean = 0;
    if (!lean)
        for (int s = 0; s < S; s++) ident[s] = false;   // the general kernel looks every key up
    for (int s = 0; s < S; s++)
        if (!((lean >> s) & 1u)) lds += steps[s].index->codec_dev.bytes();
    for (int s = 0; s < S; s++) {
        const cph_index* ix = steps[s].index;
        ChainStepArg& st = args.step[s];
        st.col = steps[s].cols[0];
        st.codec = ix->codec_dev.as<uint8_t>();
        st.codec_bytes = (int32_t)ix->codec_dev.bytes();
        if ((lean >> s) & 1u) {   // no codec block in LDS for this step
            codec_arith_plan(ix->codec, &largs.arith[s]);
            st.codec_bytes = 0;
        }
        // lookup structure of the step, built on first use (on the index's own ctx; other ctxs wait for it): the
        // 4-byte row table over a dense code space, else the hash table, else (allocation failed) the sorted codes
        st.positions = positions ? 1 : 0;
        st.rowtab = nullptr;
        st.ranktab = nullptr;
        largs.identity[s] = ident[s] ? 1 : 0;
        if (ident[s]) {
        } else if (positions) {
            CPH_TRY(index_ensure_ranktab(ctx, ix));
            st.ranktab = ix->ranktab ? ix->ranktab.as<uint2>() : nullptr;
            // in LDS when three workgroups per CU stay resident: a lean kernel keeps the plain 8-byte blocks (cheaper
            // lookups), the general one packs pairs of blocks into 12 bytes
            const size_t nblk = ranktab_blocks(ix->table_entries) / 2, rb = lean ? nblk * 16 : (nblk * 12 + 15) & ~(size_t)15;
            if (st.ranktab && ctx->chain_rank_lds && lds + rank_lds_bytes + rb <= 52 * 1024) {
                st.ranktab_lds = (int32_t)nblk;
                rank_lds_bytes += rb;
            }
        } else {
            CPH_TRY(index_ensure_rowtab(ctx, ix));
            st.rowtab = ix->rowtab ? ix->rowtab.as<uint32_t>() : nullptr;
        }
        st.src = steps[s].source;
        st.src_perm = (steps[s].source > 0 && positions) ? steps[steps[s].source - 1].index->perm.as<uint32_t>() : nullptr;
        st.hash = nullptr;
        st.hash_sectors = 0;
        st.hash_slice_mask = 0;
        if (!st.rowtab && !st.ranktab && !ident[s] && index_wants_hash(ix)) {
            CPH_TRY(index_ensure_hash(ctx, ix));
            if (ix->hash_mode == kHashK1) {
                st.hash = ix->hash.as<uint4>();
                st.hash_sectors = ix->hash_sectors;
                st.hash_slice_mask = ix->hash_slice_mask;
            }
        }
        st.codes = ix->sorted_codes.get();
        st.perm = ix->perm.as<uint32_t>();
        st.n_index = ix->nrows;
        st.key32 = ix->codec.key32 ? 1 : 0;
        args.out_rows[s] = d_rows[s];
    }
    lds += rank_lds_bytes;
    args.nt_streams = ctx->chain_nt_streams == 1 || (ctx->chain_nt_streams == 2 && positions) ? 1 : 0;
    const uint64_t ncounts = ntiles * kChainWaves;   // one match count per (tile, wave), tile-major
    using KernelFn = void (*)(ChainArgs, uint64_t, uint64_t, uint64_t, uint64_t*, uint32_t*, int, LeanArgs);
    static const KernelFn variants[2][2][2] = {
        {{&k_chain_dense<S, false, false, false>, &k_chain_dense<S, false, false, true>},
         {&k_chain_dense<S, false, true, false>, &k_chain_dense<S, false, true, true>}},
        {{&k_chain_dense<S, true, false, false>, &k_chain_dense<S, true, false, true>},
         {&k_chain_dense<S, true, true, false>, &k_chain_dense<S, true, true, true>}}};
    KernelFn kernel = variants[long_keys ? 1 : 0][wide ? 1 : 0][(dbg && !dep) ? 1 : 0];
    // register-heavy instantiations (tools/kernel_usage.sh: 256 VGPRs + AGPR copies at 8 rows per lane = ONE wave per SIMD) walk the
    // wave's rows 4 at a time: long keys with 64-bit codes from two steps on, any long or wide chain from three steps on
    if (!dbg && ctx->chain_rows4 != 0 && ((S == 2 && long_keys && wide) || (S >= 3 && (long_keys || wide)) || S == 4 || ctx->chain_rows4 == 2)) {
        static const KernelFn r4[2][2] = {{&k_chain_dense<S, false, false, false, 0u, 4>, &k_chain_dense<S, false, true, false, 0u, 4>},
                                          {&k_chain_dense<S, true, false, false, 0u, 4>, &k_chain_dense<S, true, true, false, 0u, 4>}};
        kernel = r4[long_keys ? 1 : 0][wide ? 1 : 0];
    }
    if constexpr (S >= 2) {
        if (dep) kernel = &k_chain_dense<S, true, true, false, 0u, 4, true>;
    }
    if (lean) {
        constexpr uint32_t kAll = (1u << S) - 1u;
        if (lean == kAll) kernel = &k_chain_dense<S, false, false, false, kAll>;
        if constexpr (S == 2) {
            if (lean == 1u) kernel = &k_chain_dense<S, false, false, false, 1u>;
            if (lean == 2u) kernel = &k_chain_dense<S, false, false, false, 2u>;
        }
    }
    // persistent workgroups: exactly as many as are resident at once (no tail wave), each walking
    // tiles blockIdx, blockIdx + grid, ...
    int per_cu = 1, cus = 256;
    CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(kernel), kChainThreads, lds, &per_cu));
    CPH_TRY(device_cus(ctx, &cus));
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)cus * (uint64_t)per_cu);
    {
        ProfScope ps(ctx, prof_name, 0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kChainThreads), lds, ctx->stream, args, nprobe, probe_base,
                           ntiles, d_masks, d_counts, dep ? 0 : dbg, largs);
    }
    if (host_total) {   // the total goes straight into a report word of the host (d_total unused)
        DevBuf& acc = ctx->self_clean[ctx->stream_slot].sum;
        CPH_TRY(self_clean_block(ctx, &acc, 16));
        ProfScope ps(ctx, "k_sum_counts", 4.0 * (double)ncounts);
        const unsigned sgrid = (unsigned)std::min<uint64_t>((ncounts + 255) / 256, 512);
        hipLaunchKernelGGL(k_sum_counts_report, dim3(sgrid), dim3(256), 0, ctx->stream, d_counts, ncounts, acc.as<unsigned long long>(),
                           acc.as<uint32_t>() + 2, reinterpret_cast<unsigned long long*>(host_total));
    } else if (d_total) {   // (neither: the caller adds passes that change the counts and sums them itself — run_prejoined)
        ProfScope ps(ctx, "k_sum_counts", 4.0 * (double)ncounts);
        CPH_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream));
        const unsigned sgrid = (unsigned)std::min<uint64_t>((ncounts + 255) / 256, 512);
        hipLaunchKernelGGL(k_sum_counts, dim3(sgrid), dim3(256), 0, ctx->stream, d_counts, ncounts,
                           reinterpret_cast<unsigned long long*>(d_total));
    }
    CPH_HIP_TRY(hipGetLastError());
    if (args_out) *args_out = args;
    if (grid_out) *grid_out = grid;
    return {};
}

bool chain_fast_path_ok(const ChainStep* steps, int nsteps);

Status chain_enqueue_dense(cph_ctx* ctx, const ChainStep* steps, int nsteps, uint64_t nprobe, uint64_t probe_base,
                           uint32_t* const* d_rows, uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total, bool positions) {
    switch (nsteps) {
    case 1: return enqueue_dense<1>(ctx, steps, nprobe, probe_base, d_rows, d_masks, d_counts, d_total, nullptr, nullptr, positions);
    case 2: return enqueue_dense<2>(ctx, steps, nprobe, probe_base, d_rows, d_masks, d_counts, d_total, nullptr, nullptr, positions);
    case 3: return enqueue_dense<3>(ctx, steps, nprobe, probe_base, d_rows, d_masks, d_counts, d_total, nullptr, nullptr, positions);
    case 4: return enqueue_dense<4>(ctx, steps, nprobe, probe_base, d_rows, d_masks, d_counts, d_total, nullptr, nullptr, positions);
    }
    return {CPH_ERR_INVALID, "bad chain length"};
}
// d_codes[s] = the chunk's host-formed codes for step s, already in device memory.  Every index needs a direct lookup over
// its code space: the rank table (or nothing, when the index fills its code space) for positions, the row table otherwise.
template <int S>
static Status enqueue_codes(cph_ctx* ctx, const cph_index* const* idx, const uint32_t* const* d_codes, uint64_t nprobe, uint32_t* const* d_rows,
                            uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total, bool positions) {
    const uint64_t ntiles = (nprobe + kChainTile - 1) / kChainTile;
    CodeArgs args{};
    for (int s = 0; s < S; s++) {
        const cph_index* ix = idx[s];
        if (ix->nkeycols < 1 || ix->codec.nwords != 1 || !ix->codec.key32 || !ix->windows.empty() || ix->first_dup != UINT64_MAX)
            return {CPH_ERR_INVALID, "joining host-formed codes needs duplicate-free indexes with one-word 32-bit codes"};
        CodeStepArg& st = args.step[s];
        st.codes = d_codes[s];
        st.n_index = ix->nrows;
        st.table_entries = ix->table_entries;
        if (positions && ix->table_entries != 0 && ix->table_entries == ix->nrows) {
            st.identity = 1;
        } else if (positions) {
            CPH_TRY(index_ensure_ranktab(ctx, ix));
            st.ranktab = ix->ranktab ? ix->ranktab.as<uint2>() : nullptr;
        } else {
            CPH_TRY(index_ensure_rowtab(ctx, ix));
            st.rowtab = ix->rowtab ? ix->rowtab.as<uint32_t>() : nullptr;
        }
        if (!st.identity && !st.ranktab && !st.rowtab)
            return {CPH_ERR_INVALID, "joining host-formed codes needs a direct lookup structure (dense code space) for every index"};
        args.out_rows[s] = d_rows[s];
    }
    int per_cu = 1, cus = 256;
    CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_chain_codes<S>), kChainThreads, 0, &per_cu));
    CPH_TRY(device_cus(ctx, &cus));
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)cus * (uint64_t)per_cu);
    {
        ProfScope ps(ctx, "k_chain_codes", (double)nprobe * 8.0 * S);
        hipLaunchKernelGGL(k_chain_codes<S>, dim3(grid), dim3(kChainThreads), 0, ctx->stream, args, nprobe, ntiles, d_masks, d_counts);
    }
    const uint64_t ncounts = ntiles * kChainWaves;
    {
        ProfScope ps(ctx, "k_sum_counts", 4.0 * (double)ncounts);
        CPH_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream));
        const unsigned sgrid = (unsigned)std::min<uint64_t>((ncounts + 255) / 256, 512);
        hipLaunchKernelGGL(k_sum_counts, dim3(sgrid), dim3(256), 0, ctx->stream, d_counts, ncounts, reinterpret_cast<unsigned long long*>(d_total));
    }
    CPH_HIP_TRY(hipGetLastError());
    return {};
}
Status chain_enqueue_codes(cph_ctx* ctx, const cph_index* const* idx, const uint32_t* const* d_codes, int nsteps, uint64_t nprobe,
                           uint32_t* const* d_rows, uint64_t* d_masks, uint32_t* d_counts, uint64_t* d_total, bool positions) {
    switch (nsteps) {
    case 1: return enqueue_codes<1>(ctx, idx, d_codes, nprobe, d_rows, d_masks, d_counts, d_total, positions);
    case 2: return enqueue_codes<2>(ctx, idx, d_codes, nprobe, d_rows, d_masks, d_counts, d_total, positions);
    case 3: return enqueue_codes<3>(ctx, idx, d_codes, nprobe, d_rows, d_masks, d_counts, d_total, positions);
    case 4: return enqueue_codes<4>(ctx, idx, d_codes, nprobe, d_rows, d_masks, d_counts, d_total, positions);
    }
    return {CPH_ERR_INVALID, "bad chain length"};
}
uint64_t chain_dense_mask_words(uint64_t nprobe) { return (nprobe + kChainTile - 1) / kChainTile * kChainMasks; }
uint64_t chain_dense_count_words(uint64_t nprobe) { return (nprobe + kChainTile - 1) / kChainTile * kChainWaves; }

// The chain's build sides joined with each other first: for every step whose key is a column of an earlier step's build table,
// tab[s][h] = the row (position) that step finds for the table row with handle h — one dense pass over that TABLE's column (1e7
// rows instead of 1e8 stream rows), brought into the source index's sorted order when the chain reports positions.  Then the
// stream rows take the stream-keyed steps through k_chain_dense (its lean variants included) and k_chain_prejoined answers the
// rest with one gather each; the total is summed behind it.  (people.Join(orders, "id").Join(products): csvplus_test.go:280-285.)
static Status run_prejoined(cph_ctx* ctx, const ChainStep* steps, int S, uint64_t nprobe, uint64_t probe_base, uint32_t* const* rows,
                            uint64_t* masks, uint32_t* counts, unsigned* grid_out, bool positions, uint64_t* host_total) {
    const uint64_t ntiles = (nprobe + kChainTile - 1) / kChainTile;
    const uint64_t ncounts = ntiles * kChainWaves;
    DevBuf tabs[kMaxChain];
    for (int s = 0; s < S; s++) {
        if (steps[s].source == 0) continue;
        const int t = (steps[s].source < 0 ? -steps[s].source : steps[s].source) - 1;
        const uint64_t nt = steps[t].index->nrows;
        ChainStep one = steps[s];
        one.source = 0;
        DevBuf trow, tmask, tcount;
        CPH_TRY(trow.alloc(&ctx->pool, nt * sizeof(uint32_t)));
        CPH_TRY(tmask.alloc(&ctx->pool, chain_dense_mask_words(nt) * sizeof(uint64_t)));
        CPH_TRY(tcount.alloc(&ctx->pool, chain_dense_count_words(nt) * sizeof(uint32_t)));
        CPH_HIP_TRY(hipMemsetAsync(trow.get(), 0xFF, nt * sizeof(uint32_t), ctx->stream));   // the dense pass stores matches only
        uint32_t* r1[kMaxChain] = {trow.as<uint32_t>(), nullptr, nullptr, nullptr};
        CPH_TRY(enqueue_dense<1>(ctx, &one, nt, 0, r1, tmask.as<uint64_t>(), tcount.as<uint32_t>(), nullptr, nullptr, nullptr, positions, nullptr,
                                 "k_chain_prejoin_table"));
        if (positions && steps[s].source > 0) {   // the column is in the table's row order, the chain hands sorted positions on
            CPH_TRY(tabs[s].alloc(&ctx->pool, nt * sizeof(uint32_t)));
            ProfScope ps(ctx, "k_chain_prejoin_table", 0);
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)std::min<uint64_t>((nt + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                               trow.as<uint32_t>(), steps[t].index->perm.as<uint32_t>(), tabs[s].as<uint32_t>(), nt);
            CPH_HIP_TRY(hipGetLastError());
        } else {
            tabs[s] = std::move(trow);
        }
    }
    ChainStep ind[kMaxChain];
    uint32_t* irows[kMaxChain] = {nullptr, nullptr, nullptr, nullptr};
    int ni = 0;
    PrejoinArgs pa{};
    int dep_of[kMaxChain] = {-1, -1, -1, -1};
    for (int s = 0; s < S; s++) {
        if (steps[s].source == 0) {
            ind[ni] = steps[s];
            irows[ni++] = rows[s];
            continue;
        }
        const int t = (steps[s].source < 0 ? -steps[s].source : steps[s].source) - 1;
        const int j = pa.n++;
        dep_of[s] = j;
        pa.src_dep[j] = dep_of[t];
        pa.src_rows[j] = rows[t];
        pa.tab[j] = tabs[s].as<uint32_t>();
        pa.out_rows[j] = rows[s];
    }
    switch (ni) {
    case 1: CPH_TRY(enqueue_dense<1>(ctx, ind, nprobe, probe_base, irows, masks, counts, nullptr, nullptr, grid_out, positions)); break;
    case 2: CPH_TRY(enqueue_dense<2>(ctx, ind, nprobe, probe_base, irows, masks, counts, nullptr, nullptr, grid_out, positions)); break;
    case 3: CPH_TRY(enqueue_dense<3>(ctx, ind, nprobe, probe_base, irows, masks, counts, nullptr, nullptr, grid_out, positions)); break;
    default: return {CPH_ERR_INVALID, "a chain needs a step keyed by the stream"};
    }
    {
        int per_cu = 1, cus = 256;
        CPH_TRY(kernel_setup(ctx, reinterpret_cast<const void*>(&k_chain_prejoined), kChainThreads, 0, &per_cu));
        CPH_TRY(device_cus(ctx, &cus));
        const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)cus * (uint64_t)per_cu);
        ProfScope ps(ctx, "k_chain_prejoined", (double)nprobe * 8.0 * pa.n);
        hipLaunchKernelGGL(k_chain_prejoined, dim3(grid), dim3(kChainThreads), 0, ctx->stream, pa, nprobe, ntiles, masks, counts);
        CPH_HIP_TRY(hipGetLastError());
    }
    {
        DevBuf& acc = ctx->self_clean[ctx->stream_slot].sum;
        CPH_TRY(self_clean_block(ctx, &acc, 16));
        ProfScope ps(ctx, "k_sum_counts", 4.0 * (double)ncounts);
        const unsigned sgrid = (unsigned)std::min<uint64_t>((ncounts + 255) / 256, 512);
        hipLaunchKernelGGL(k_sum_counts_report, dim3(sgrid), dim3(256), 0, ctx->stream, counts, ncounts, acc.as<unsigned long long>(),
                           acc.as<uint32_t>() + 2, reinterpret_cast<unsigned long long*>(host_total));
        CPH_HIP_TRY(hipGetLastError());
    }
    // the tables are released here: stream-ordered reuse keeps them alive for the kernels already enqueued
    return {};
}

template <int S>
static Status run_fast(cph_ctx* ctx, const ChainStep* steps, uint64_t nprobe, uint64_t probe_base, ChainOut* out, bool positions) {
    const uint64_t ntiles = (nprobe + kChainTile - 1) / kChainTile;
    const uint64_t ncounts = ntiles * kChainWaves;
    uint32_t* rows[kMaxChain] = {nullptr};
    for (int s = 0; s < S; s++) {
        CPH_TRY(out->build_row[s].alloc(&ctx->pool, nprobe * sizeof(uint32_t)));
        rows[s] = out->build_row[s].as<uint32_t>();
    }
    DevBuf masks, counts;
    CPH_TRY(masks.alloc(&ctx->pool, ntiles * kChainMasks * sizeof(uint64_t)));
    CPH_TRY(counts.alloc(&ctx->pool, ncounts * sizeof(uint32_t)));
    // the match total lands in a report word of the host (written by k_sum_counts_report's last workgroup): one wait, no copy
    volatile uint64_t* h = reinterpret_cast<volatile uint64_t*>(host_word(ctx, 2));
    if (!h) return {CPH_ERR_HIP, "no pinned host memory for the match total"};
    ChainArgs args{};
    unsigned grid = 1;
    // steps keyed by an earlier build table: worth pre-joining the tables when the stream is much longer than they are
    bool prejoin = ctx->chain_prejoin != 0;
    {
        bool dep = false;
        for (int s = 0; s < S; s++) {
            if (steps[s].source == 0) continue;
            dep = true;
            const int t = (steps[s].source < 0 ? -steps[s].source : steps[s].source) - 1;
            if (steps[t].index->nrows * 2 > nprobe) prejoin = false;
        }
        prejoin = prejoin && dep;
    }
    if (prejoin) {
        CPH_TRY(run_prejoined(ctx, steps, S, nprobe, probe_base, rows, masks.as<uint64_t>(), counts.as<uint32_t>(), &grid, positions,
                              const_cast<uint64_t*>(h)));
        for (int s = 0; s < S; s++) args.out_rows[s] = rows[s];   // (all the compaction reads of it)
    } else {
        CPH_TRY(enqueue_dense<S>(ctx, steps, nprobe, probe_base, rows, masks.as<uint64_t>(), counts.as<uint32_t>(),
                                 nullptr, &args, &grid, positions, const_cast<uint64_t*>(h)));
    }
    CPH_HIP_TRY(hipGetLastError());
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t nmatch = h[0];
    out->nrows = nmatch;
    out->identity = nmatch == nprobe;   // result row m is stream row probe_base + m: no stream_row array
    if (nmatch == nprobe || nmatch == 0) return {};   // dense arrays are already final / nothing to keep

    // compaction: tile bases, then move the tuples
    CPH_TRY(exclusive_scan_u32(ctx, counts.as<uint32_t>(), ncounts));
    ChainOut fin;
    CPH_TRY(fin.stream_row.alloc(&ctx->pool, nmatch * sizeof(uint64_t)));
    ChainArgs fargs{};
    for (int s = 0; s < S; s++) {
        CPH_TRY(fin.build_row[s].alloc(&ctx->pool, nmatch * sizeof(uint32_t)));
        fargs.out_rows[s] = fin.build_row[s].as<uint32_t>();
    }
    {
        ProfScope ps(ctx, "k_chain_compact", 2.0 * (double)nmatch * (8.0 + 4.0 * S));
        hipLaunchKernelGGL(k_chain_compact<S>, dim3(grid), dim3(kChainThreads), 0, ctx->stream, args, probe_base,
                           masks.as<uint64_t>(), counts.as<uint32_t>(), ntiles,
                           fin.stream_row.as<uint64_t>(), fargs);
    }
    CPH_HIP_TRY(hipGetLastError());
    out->stream_row = std::move(fin.stream_row);
    for (int s = 0; s < S; s++) out->build_row[s] = std::move(fin.build_row[s]);
    return {};
}

// One step of the general path through a pre-joined table (k_prejoin_tuples / k_prejoin_compact above).  h = the row of table t
// every tuple reads its key from (the handle the step's columns are indexed by); *n tuples in, *n out.
static Status prejoin_general_step(cph_ctx* ctx, const ChainStep* steps, int s, int t, const uint32_t* h, const uint32_t* h_perm, uint64_t n_in, bool positions,
                                   DevBuf* cur_stream, DevBuf* cur_rows, uint64_t* n_out, bool* all) {
    const uint64_t nt = steps[s].cols[0].nrows;
    // the table pass: tab[row of table t] = the sorted position / build row this step finds for that row's key, or absent
    ChainStep one = steps[s];
    one.source = 0;
    DevBuf tab, tmask, tcount;
    CPH_TRY(tab.alloc(&ctx->pool, nt * sizeof(uint32_t)));
    CPH_TRY(tmask.alloc(&ctx->pool, chain_dense_mask_words(nt) * sizeof(uint64_t)));
    CPH_TRY(tcount.alloc(&ctx->pool, chain_dense_count_words(nt) * sizeof(uint32_t)));
    CPH_HIP_TRY(hipMemsetAsync(tab.get(), 0xFF, nt * sizeof(uint32_t), ctx->stream));   // the dense pass stores matches only
    uint32_t* r1[kMaxChain] = {tab.as<uint32_t>(), nullptr, nullptr, nullptr};
    CPH_TRY(enqueue_dense<1>(ctx, &one, nt, 0, r1, tmask.as<uint64_t>(), tcount.as<uint32_t>(), nullptr, nullptr, nullptr, positions, nullptr,
                             "k_chain_prejoin_table"));
    const uint64_t ntiles = (n_in + kPjTile - 1) / kPjTile;
    DevBuf v, counts;
    CPH_TRY(v.alloc(&ctx->pool, n_in * sizeof(uint32_t)));
    CPH_TRY(counts.alloc(&ctx->pool, (ntiles + 1) * sizeof(uint32_t)));
    volatile uint64_t* host_total = reinterpret_cast<volatile uint64_t*>(host_word(ctx, 2));
    if (!host_total) return {CPH_ERR_HIP, "no pinned host memory for the match total"};
    {
        ProfScope ps(ctx, "k_prejoin_tuples", (double)n_in * (h_perm ? 16.0 : 12.0));
        hipLaunchKernelGGL(k_prejoin_tuples, dim3((unsigned)ntiles), dim3(kPjThreads), 0, ctx->stream, tab.as<uint32_t>(), h, h_perm, n_in, v.as<uint32_t>(),
                           counts.as<uint32_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    {
        DevBuf& acc = ctx->self_clean[ctx->stream_slot].sum;
        CPH_TRY(self_clean_block(ctx, &acc, 16));
        ProfScope ps(ctx, "k_sum_counts", 4.0 * (double)ntiles);
        const unsigned sgrid = (unsigned)std::min<uint64_t>((ntiles + 255) / 256, 512);
        hipLaunchKernelGGL(k_sum_counts_report, dim3(sgrid), dim3(256), 0, ctx->stream, counts.as<uint32_t>(), ntiles, acc.as<unsigned long long>(),
                           acc.as<uint32_t>() + 2, reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(host_total)));
        CPH_HIP_TRY(hipGetLastError());
    }
    CPH_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t m = host_total[0];
    *all = m == n_in;
    *n_out = m;
    if (m == n_in) {   // every tuple survived: the step only adds its column
        cur_rows[s] = std::move(v);
        return {};
    }
    if (m == 0) return {};
    CPH_TRY(exclusive_scan_u32(ctx, counts.as<uint32_t>(), ntiles));
    PjCompactArgs ca{};
    DevBuf nstream, nrows[CPH_MAX_CHAIN];
    CPH_TRY(nstream.alloc(&ctx->pool, m * sizeof(uint64_t)));
    ca.src64 = cur_stream->as<uint64_t>();
    ca.dst64 = nstream.as<uint64_t>();
    for (int u = 0; u < s; u++) {
        CPH_TRY(nrows[u].alloc(&ctx->pool, m * sizeof(uint32_t)));
        ca.src32[ca.n32] = cur_rows[u].as<uint32_t>();
        ca.dst32[ca.n32++] = nrows[u].as<uint32_t>();
    }
    CPH_TRY(nrows[s].alloc(&ctx->pool, m * sizeof(uint32_t)));
    ca.src32[ca.n32] = v.as<uint32_t>();
    ca.dst32[ca.n32++] = nrows[s].as<uint32_t>();
    {
        ProfScope ps(ctx, "k_compose", (double)n_in * (12.0 + 4.0 * s) + (double)m * (12.0 + 4.0 * s));
        hipLaunchKernelGGL(k_prejoin_compact, dim3((unsigned)ntiles), dim3(kPjThreads), 0, ctx->stream, ca, v.as<uint32_t>(), n_in, counts.as<uint32_t>());
        CPH_HIP_TRY(hipGetLastError());
    }
    *cur_stream = std::move(nstream);
    for (int u = 0; u <= s; u++) cur_rows[u] = std::move(nrows[u]);
    return {};
}

Status chain_run(cph_ctx* ctx, const ChainStep* steps, int nsteps, uint64_t probe_base, ChainOut* out, bool positions) {
    const uint64_t nprobe = steps[0].cols[0].nrows;
    out->nrows = 0;
    out->nsteps = nsteps;
    if (nprobe == 0) return {};
    for (int s = 0; s < nsteps; s++)
        if (steps[s].index->nrows == 0) return {};   // a Join with an empty index emits nothing (csvplus.go:557-559: the loop body never runs)

    if (chain_fast_path_ok(steps, nsteps)) {
        size_t lds = 0;
        for (int s = 0; s < nsteps; s++) lds += steps[s].index->codec_dev.bytes();
        if (lds <= 150 * 1024) {
            switch (nsteps) {
            case 1: return run_fast<1>(ctx, steps, nprobe, probe_base, out, positions);
            case 2: return run_fast<2>(ctx, steps, nprobe, probe_base, out, positions);
            case 3: return run_fast<3>(ctx, steps, nprobe, probe_base, out, positions);
            default: return run_fast<4>(ctx, steps, nprobe, probe_base, out, positions);
            }
        }
    }

    // ---- general path: probe / select / compose, one step at a time ------------------------------------
    ProbeOut first;
    CPH_TRY(probe_run(ctx, steps[0].index, steps[0].cols, steps[0].ncols, RowSel{}, nprobe, probe_base, true, &first, positions));
    DevBuf cur_stream = std::move(first.pidx);
    DevBuf cur_rows[CPH_MAX_CHAIN];
    cur_rows[0] = std::move(first.brow);
    uint64_t n = first.nmatches;
    for (int s = 1; s < nsteps && n > 0; s++) {
        // the rows this step reads its key from: the stream rows of the tuples so far, or the rows an earlier step matched
        // in that step's build table (cph_chain_step.source; mergeRows put that row's columns into the row this Join sees)
        RowSel sel;
        DevBuf src_rows;
        // (round 6) a duplicate-free single-column index keyed by an earlier build table whose rows are not many more than the tuples:
        // join the TABLE with this index once (dense pass), then one 4-byte gather per tuple
        if (steps[s].source != 0 && ctx->chain_prejoin != 0 && n < (1ull << 32) && chain_fast_path_ok(&steps[s], 1) &&
            steps[s].index->codec_dev.bytes() <= 150 * 1024) {
            const int t = (steps[s].source < 0 ? -steps[s].source : steps[s].source) - 1;
            const uint64_t nt = steps[s].cols[0].nrows;
            if (nt <= 2 * n && nt == steps[t].index->nrows) {
                bool survived_all = false;
                CPH_TRY(prejoin_general_step(ctx, steps, s, t, cur_rows[t].as<uint32_t>(),
                                             steps[s].source > 0 && positions ? steps[t].index->perm.as<uint32_t>() : nullptr, n, positions,
                                             &cur_stream, cur_rows, &n, &survived_all));
                continue;
            }
        }
        if (steps[s].source == 0) {
            sel.ptr = cur_stream.get();
            sel.bits = 64;
            sel.base = probe_base;
        } else {
            const int t = (steps[s].source < 0 ? -steps[s].source : steps[s].source) - 1;
            sel.ptr = cur_rows[t].get();
            sel.bits = 32;
            sel.base = 0;
            if (steps[s].source > 0 && positions) {   // columns in the table's original order, tuples hold sorted positions
                CPH_TRY(src_rows.alloc(&ctx->pool, n * sizeof(uint32_t)));
                ProfScope ps(ctx, "k_compose", (double)n * 12.0);
                hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                                   steps[t].index->perm.as<uint32_t>(), cur_rows[t].as<uint32_t>(), src_rows.as<uint32_t>(), n);
                CPH_HIP_TRY(hipGetLastError());
                sel.ptr = src_rows.get();
            }
        }
        ProbeOut po;
        CPH_TRY(probe_run(ctx, steps[s].index, steps[s].cols, steps[s].ncols, sel, n, 0, true, &po, positions));
        const uint64_t m = po.nmatches;
        DevBuf nstream;
        CPH_TRY(nstream.alloc(&ctx->pool, m * sizeof(uint64_t)));
        unsigned grid = (unsigned)std::min<uint64_t>((m + 255) / 256, 8192);
        if (m) {
            ProfScope ps(ctx, "k_compose", (double)m * (8.0 + 16.0 + 12.0 * s));
            hipLaunchKernelGGL(k_compose_u64, dim3(grid), dim3(256), 0, ctx->stream, cur_stream.as<uint64_t>(),
                               po.pidx.as<uint64_t>(), nstream.as<uint64_t>(), m);
            for (int t = 0; t < s; t++) {
                DevBuf nr;
                CPH_TRY(nr.alloc(&ctx->pool, m * sizeof(uint32_t)));
                hipLaunchKernelGGL(k_compose_u32, dim3(grid), dim3(256), 0, ctx->stream, cur_rows[t].as<uint32_t>(),
                                   po.pidx.as<uint64_t>(), nr.as<uint32_t>(), m);
                cur_rows[t] = std::move(nr);
            }
            CPH_HIP_TRY(hipGetLastError());
        }
        cur_stream = std::move(nstream);
        cur_rows[s] = std::move(po.brow);
        n = m;
    }
    out->nrows = n;
    out->stream_row = std::move(cur_stream);
    for (int s = 0; s < nsteps; s++) out->build_row[s] = std::move(cur_rows[s]);
    return {};
}

}  // namespace cph

// Loads this translation unit's code object now (cph_ctx_create) instead of inside the first timed call.
namespace cph {
void warm_chain() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sum_counts));
    (void)hipGetLastError();
}
}  // namespace cph

// hash_device.hpp — the hash probe of Join for code spaces too sparse for a direct-address table.
//
// The reference's first() (csvplus.go:893-897) is a binary search whose cost does not depend on what the keys look
// like.  The device path has a direct-address table when the key codes are dense (probe.hip: index_plan_table) —
// one load per probe row — and, without it, fell back to a per-row binary search over the sorted codes: ~24
// dependent, divergent loads per row at 1e7 index rows, six times slower.  This table closes that gap for ANY key
// (random ids, hashes, multi-word codes, keys of several codec windows):
//
//   * open addressing over SECTORS of 64 bytes (what one random access moves from HBM / the Infinity Cache anyway);
//     a key's home sector = mulhi(hash >> 32, nsectors); a sector holds 4 entries of 16 bytes (or 2 of 32 bytes)
//   * one entry per DISTINCT key, inserted at the first empty slot of the probe sequence home, home + 1, ...
//     (slots in sector order); entries are never removed, so a lookup may stop at the first sector that still has an
//     empty slot — at load factor 0.5 that is the home sector for ~95 % of the keys: ONE sector per probe row.  The
//     load factor counts DISTINCT keys since round 5 (ctx option hash_load_pct, default 50: denser tables measured
//     slower — 75 %: 3.58 ms against 2.69 ms for 1e8 probes of 1e7 keys, the longer probe sequences cost more than
//     the smaller table saves).  Following full sectors in ROUNDS over a lane's 4 rows instead of row by row was
//     measured too: 3.23 ms against 2.69 (the wave-wide loop costs more than the few lanes that continue)
//   * an entry carries the key itself when it fits — so a hit needs no second access to verify — and what Join
//     wants to know:  lo = sorted position of the key's first row, aux = the build row perm[lo] (index without
//     duplicate keys: no dependent perm gather) or the end of the key's run of rows (index with duplicates)
//
// Three entry formats (cph_index::hash_mode):
//   kHashK1   one code word:    {u64 code, u32 lo, u32 aux}                       exact, 16 bytes
//   kHashK3   two or three code words (up to 189 bits: UUIDs, 16 random bytes, two id columns):
//                               {u64 w0, u64 w1, u64 w2, u32 lo, u32 aux}         exact, 32 bytes
//   kHashTag  anything longer (4+ words, several key windows): {u64 tag, u32 lo, u32 aux} with tag = the 64-bit
//             hash of all words; a tag match is VERIFIED against the sorted codes at lo (word by word), so the
//             result is exact; the build makes sure no two distinct index keys share a tag (else no hash table)
// Prefix joins (fewer probe columns than the index has) need the ORDER of the codes and stay on the sorted path.
#pragma once

#include "cph_internal.hpp"
#include "device_utils.hpp"

namespace cph {

enum : int32_t { kHashNone = 0, kHashK1 = 1, kHashK3 = 2, kHashTag = 3 };
constexpr uint64_t kHashEmpty = ~0ull;            // no code word (< 2^63) and no tag (top bit cleared) equals it
constexpr uint32_t kHashAbsent = 0xFFFFFFFFu;

struct HashEntry16 {            // kHashK1 / kHashTag: 4 per sector
    uint64_t key;
    uint32_t lo, aux;
};
struct HashEntry32 {            // kHashK3: 2 per sector (w2 = 0 for two-word codes)
    uint64_t w0, w1, w2;
    uint32_t lo, aux;
};
static_assert(sizeof(HashEntry16) == 16 && sizeof(HashEntry32) == 32, "entry layout");

struct HashView {
    const uint4* sectors = nullptr;   // 4 x uint4 per sector
    uint32_t nsectors = 0;
    uint32_t slice_mask = 0;          // != 0: probe sequences wrap inside slices of slice_mask + 1 sectors (a table built slice by slice)
};

// ---- the hash -------------------------------------------------------------------------------------------------
CPH_HD inline uint64_t hash_fmix(uint64_t h) {
    h ^= h >> 33;
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
    h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 33;
    return h;
}
constexpr uint64_t kHashSeed = 0x9E3779B97F4A7C15ull;
// running state over the code words, most significant word first: s = hash_step(s, word); start with kHashSeed
CPH_HD inline uint64_t hash_step(uint64_t s, uint64_t word) { return (s ^ word) * 0x9FB21C651E98DF25ull + 0x2545F4914F6CDD1Dull; }
CPH_HD inline uint64_t hash_finish(uint64_t s) { return hash_fmix(s); }
CPH_HD inline uint64_t hash_one(uint64_t code) { return hash_finish(hash_step(kHashSeed, code)); }
CPH_HD inline uint64_t hash_tag(uint64_t h) { return h & 0x7FFFFFFFFFFFFFFFull; }
CPH_HD inline uint32_t hash_home(uint64_t h, uint32_t nsectors) { return (uint32_t)(((h >> 32) * (uint64_t)nsectors) >> 32); }
// the sector that follows s in a probe sequence
CPH_HD inline uint32_t hash_next_sector(uint32_t s, uint32_t nsectors, uint32_t slice_mask) {
    return slice_mask ? ((s & ~slice_mask) | ((s + 1u) & slice_mask)) : (s + 1u == nsectors ? 0u : s + 1u);
}

#if defined(__HIPCC__)
// ---- lookups ----------------------------------------------------------------------------------------------------
// The four 16-byte loads of a sector are issued together; the caller keeps several rows in flight by calling
// hash_load_sector for all of them first and hash_resolve_* afterwards (straight-line code: codec_device.hpp).
struct HashSector {
    uint4 e[4];
};
__device__ __forceinline__ HashSector hash_load_sector(const HashView& hv, uint32_t s) {
    const uint4* p = hv.sectors + (uint64_t)s * 4;
    HashSector r;
    r.e[0] = p[0];
    r.e[1] = p[1];
    r.e[2] = p[2];
    r.e[3] = p[3];
    return r;
}
__device__ __forceinline__ uint64_t hash_key_of(const uint4& e) { return (uint64_t)e.x | ((uint64_t)e.y << 32); }

// 16-byte entries keyed by `key` (a code word, or a tag).  Returns true on a hit (lo / aux filled); *more = the sector
// is full and does not hold the key: the search continues in the next sector (rare).
__device__ __forceinline__ bool hash_match16(const HashSector& sc, uint64_t key, uint32_t* lo, uint32_t* aux, bool* more) {
    bool hit = false, empty = false;
    uint32_t l = kHashAbsent, a = kHashAbsent;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t k = hash_key_of(sc.e[j]);
        const bool m = k == key;
        l = m ? sc.e[j].z : l;
        a = m ? sc.e[j].w : a;
        hit |= m;
        empty |= k == kHashEmpty;
    }
    *lo = l;
    *aux = a;
    *more = !hit && !empty;
    return hit;
}
// 32-byte entries keyed by up to three words
__device__ __forceinline__ bool hash_match32(const HashSector& sc, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t* lo, uint32_t* aux,
                                             bool* more) {
    bool hit = false, empty = false;
    uint32_t l = kHashAbsent, a = kHashAbsent;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint64_t k0 = hash_key_of(sc.e[2 * j]);
        const uint64_t k1 = (uint64_t)sc.e[2 * j].z | ((uint64_t)sc.e[2 * j].w << 32);
        const uint64_t k2 = hash_key_of(sc.e[2 * j + 1]);
        const bool m = k0 == w0 && k1 == w1 && k2 == w2;
        l = m ? sc.e[2 * j + 1].z : l;
        a = m ? sc.e[2 * j + 1].w : a;
        hit |= m;
        empty |= k0 == kHashEmpty;
    }
    *lo = l;
    *aux = a;
    *more = !hit && !empty;
    return hit;
}

// Complete lookups (home sector, then the following ones while they are full).
__device__ __forceinline__ bool hash_find16(const HashView& hv, uint64_t h, uint64_t key, uint32_t* lo, uint32_t* aux) {
    uint32_t s = hash_home(h, hv.nsectors);
    for (;;) {
        const HashSector sc = hash_load_sector(hv, s);
        bool more;
        if (hash_match16(sc, key, lo, aux, &more)) return true;
        if (!more) return false;
        s = hash_next_sector(s, hv.nsectors, hv.slice_mask);
    }
}
__device__ __forceinline__ bool hash_find32(const HashView& hv, uint64_t h, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t* lo, uint32_t* aux) {
    uint32_t s = hash_home(h, hv.nsectors);
    for (;;) {
        const HashSector sc = hash_load_sector(hv, s);
        bool more;
        if (hash_match32(sc, w0, w1, w2, lo, aux, &more)) return true;
        if (!more) return false;
        s = hash_next_sector(s, hv.nsectors, hv.slice_mask);
    }
}
// continues a lookup whose home sector was full (the rare tail of the straight-line callers)
__device__ __forceinline__ bool hash_continue16(const HashView& hv, uint32_t home, uint64_t key, uint32_t* lo, uint32_t* aux) {
    uint32_t s = hash_next_sector(home, hv.nsectors, hv.slice_mask);
    for (;;) {
        const HashSector sc = hash_load_sector(hv, s);
        bool more;
        if (hash_match16(sc, key, lo, aux, &more)) return true;
        if (!more) return false;
        s = hash_next_sector(s, hv.nsectors, hv.slice_mask);
    }
}
__device__ __forceinline__ bool hash_continue32(const HashView& hv, uint32_t home, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t* lo,
                                                uint32_t* aux) {
    uint32_t s = hash_next_sector(home, hv.nsectors, hv.slice_mask);
    for (;;) {
        const HashSector sc = hash_load_sector(hv, s);
        bool more;
        if (hash_match32(sc, w0, w1, w2, lo, aux, &more)) return true;
        if (!more) return false;
        s = hash_next_sector(s, hv.nsectors, hv.slice_mask);
    }
}
#endif

}  // namespace cph

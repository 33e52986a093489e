/*
 * datagen.c — deterministic synthetic table generator (CPU, OpenMP).
 *
 * Produces Arrow-style string columns (uint8 data[] + uint32/uint64
 * offsets[nrows+1]) shaped like the reference's test fixtures
 * (csvplus_test.go:1207-1333: people / stock / orders), scaled to the
 * BASELINE.json configs.  Everything is counter-based
 * (splitmix64 of (seed, row)), so any row range can be generated
 * independently: ranks generate their own probe shard, streaming chunks are
 * generated on the fly, and the result is identical for every thread count.
 *
 * This is bench/test plumbing, not part of the hot path.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>

#define DG_API __attribute__((visibility("default")))

/* ---- column kinds ------------------------------------------------------ */
enum {
    DG_SEQ_PERM  = 0, /* value = feistel_perm(row) over [0,domain): unique, unsorted ids */
    DG_UNIFORM   = 1, /* value = base + U[0,domain): foreign keys / qty             */
    DG_NAME      = 2, /* peopleNames[(id/12)%10], id = feistel_perm(row)             */
    DG_SURNAME   = 3, /* peopleSurnames[id%12]                                        */
    DG_PRODUCT   = 4, /* stockItems[id%8].name + "-" + id                             */
    DG_PRICE     = 5, /* "%.2f" of (id%8+1)/100 + (id/8 % 1000)                      */
    DG_VARKEY    = 6, /* surname "/" name "#" decimal(U[0,domain))  (config 3)        */
    DG_SEQ       = 7, /* value = row (sorted ids, for adversarial/sorted-input tests) */
    DG_UNIFORM_PERM = 8, /* value = feistel_perm(U[0,domain)) — same set as UNIFORM  */
    DG_FK_SUBSET = 9, /* value = feistel_perm(U[0,base), domain, seed): the id of a uniformly drawn row among the FIRST
                         `base` rows of the DG_SEQ_PERM column with this domain and seed — foreign keys into a table
                         whose ids occupy only part of their id space (domain > rows)                                */
    DG_RANDKEY   = 10 /* 12 characters [a-z0-9]: the base-36 digits of a 62-bit bijection of id, id = row (base == 0:
                         distinct keys of a build table) or U[0,base) (foreign keys into its first `base` rows)      */
};
enum { DG_ITOA = 0, DG_FIXED8 = 1 };

typedef struct {
    int32_t  kind;
    int32_t  encoding;   /* DG_ITOA | DG_FIXED8 (numeric kinds only) */
    uint64_t domain;     /* value domain size                         */
    uint64_t base;       /* added to UNIFORM values (qty: base=1)     */
    uint64_t seed;
} dg_spec;

static const char* const kNames[10] = {
    "Amelia", "Olivia", "Emily", "Ava", "Isla",
    "Oliver", "Jack", "Harry", "Jacob", "Charlie",
};
static const char* const kSurnames[12] = {
    "Smith", "Jones", "Taylor", "Williams", "Brown", "Davies",
    "Evans", "Wilson", "Thomas", "Roberts", "Johnson", "Lewis",
};
static const char* const kStock[8] = {
    "banana", "apple", "orange", "pea", "tomato", "potato", "cucumber", "iPhone",
};

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline uint64_t rnd(uint64_t seed, uint64_t row, uint64_t stream) {
    return splitmix64(splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + row);
}

/* unbiased-enough bounded draw: 64x64->128 multiply-shift */
static inline uint64_t bounded(uint64_t r, uint64_t n) {
    return (uint64_t)(((unsigned __int128)r * n) >> 64);
}

/* Bijection on [0,n): 4-round Feistel over the enclosing power-of-4 domain
 * with cycle walking. */
static uint64_t feistel_perm(uint64_t x, uint64_t n, uint64_t seed) {
    if (n <= 1) return 0;
    if (x >= n) x %= n;   /* rows past the domain wrap around (cycle walking would never return for them) */
    int bits = 0;
    while (((uint64_t)1 << bits) < n) bits++;
    if (bits & 1) bits++;
    const int half = bits / 2;
    const uint64_t mask = ((uint64_t)1 << half) - 1;
    do {
        uint64_t l = x >> half, r = x & mask;
        for (int round = 0; round < 4; round++) {
            uint64_t f = splitmix64(r ^ splitmix64(seed + (uint64_t)round)) & mask;
            uint64_t t = l ^ f;
            l = r;
            r = t;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

static inline int fmt_u64(uint64_t v, int encoding, char* out) {
    if (encoding == DG_FIXED8) {
        for (int i = 7; i >= 0; i--) { out[i] = (char)('0' + v % 10); v /= 10; }
        return 8;
    }
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return n;
}

/* Formats row `row` of the column into out (>= 64 bytes); returns length. */
static int gen_value(const dg_spec* s, uint64_t row, char* out) {
    switch (s->kind) {
    case DG_SEQ_PERM:
        return fmt_u64(feistel_perm(row, s->domain, s->seed), s->encoding, out);
    case DG_SEQ:
        return fmt_u64(row, s->encoding, out);
    case DG_UNIFORM:
        return fmt_u64(s->base + bounded(rnd(s->seed, row, 1), s->domain), s->encoding, out);
    case DG_UNIFORM_PERM:
        return fmt_u64(s->base + feistel_perm(bounded(rnd(s->seed, row, 1), s->domain), s->domain, s->seed ^ 0x51ED),
                       s->encoding, out);
    case DG_FK_SUBSET:
        return fmt_u64(feistel_perm(bounded(rnd(s->seed ^ 0xF00Dull, row, 1), s->base), s->domain, s->seed), s->encoding, out);
    case DG_RANDKEY: {
        static const char kA36[] = "abcdefghijklmnopqrstuvwxyz0123456789";
        uint64_t id = s->base ? bounded(rnd(s->seed ^ 0xF00Dull, row, 1), s->base) : row;
        uint64_t v = feistel_perm(id, (uint64_t)1 << 62, s->seed);   /* < 2^62 < 36^12 */
        for (int i = 11; i >= 0; i--) { out[i] = kA36[v % 36]; v /= 36; }
        return 12;
    }
    case DG_NAME: {
        uint64_t id = feistel_perm(row, s->domain, s->seed);
        const char* p = kNames[(id / 12) % 10];
        int n = (int)strlen(p);
        memcpy(out, p, (size_t)n);
        return n;
    }
    case DG_SURNAME: {
        uint64_t id = feistel_perm(row, s->domain, s->seed);
        const char* p = kSurnames[id % 12];
        int n = (int)strlen(p);
        memcpy(out, p, (size_t)n);
        return n;
    }
    case DG_PRODUCT: {
        uint64_t id = feistel_perm(row, s->domain, s->seed);
        const char* p = kStock[id % 8];
        int n = (int)strlen(p);
        memcpy(out, p, (size_t)n);
        out[n++] = '-';
        return n + fmt_u64(id, DG_ITOA, out + n);
    }
    case DG_PRICE: {
        uint64_t id = feistel_perm(row, s->domain, s->seed);
        uint64_t cents = (id % 8 + 1) + ((id / 8) % 1000) * 100;
        int n = fmt_u64(cents / 100, DG_ITOA, out);
        out[n++] = '.';
        out[n++] = (char)('0' + (cents / 10) % 10);
        out[n++] = (char)('0' + cents % 10);
        return n;
    }
    case DG_VARKEY: {
        uint64_t r = rnd(s->seed, row, 2);
        const char* sn = kSurnames[r % 12];
        const char* nm = kNames[(r / 12) % 10];
        int n = (int)strlen(sn);
        memcpy(out, sn, (size_t)n);
        out[n++] = '/';
        int m = (int)strlen(nm);
        memcpy(out + n, nm, (size_t)m);
        n += m;
        out[n++] = '#';
        return n + fmt_u64(bounded(rnd(s->seed, row, 3), s->domain), DG_ITOA, out + n);
    }
    default:
        return 0;
    }
}

/* Total data bytes of rows [row0, row0+nrows). */
DG_API uint64_t dg_column_bytes(const dg_spec* s, uint64_t row0, uint64_t nrows) {
    if (s->encoding == DG_FIXED8 &&
        (s->kind == DG_SEQ_PERM || s->kind == DG_UNIFORM || s->kind == DG_SEQ || s->kind == DG_UNIFORM_PERM || s->kind == DG_FK_SUBSET))
        return nrows * 8;
    if (s->kind == DG_RANDKEY) return nrows * 12;
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static)
    for (int64_t i = 0; i < (int64_t)nrows; i++) {
        char buf[64];
        total += (uint64_t)gen_value(s, row0 + (uint64_t)i, buf);
    }
    return total;
}

/*
 * Fills data[] and offsets[nrows+1] (offset_bits = 32 or 64) for rows
 * [row0, row0+nrows).  Returns total bytes, or UINT64_MAX if the data would
 * not fit 32-bit offsets.
 */
DG_API uint64_t dg_column_fill(const dg_spec* s, uint64_t row0, uint64_t nrows, uint8_t* data, void* offsets,
                               int32_t offset_bits) {
    uint32_t* o32 = offset_bits == 32 ? (uint32_t*)offsets : NULL;
    uint64_t* o64 = offset_bits == 64 ? (uint64_t*)offsets : NULL;
    enum { CHUNK = 1 << 16 };
    const uint64_t nchunks = (nrows + CHUNK - 1) / CHUNK;
    uint64_t* chunk_bytes = NULL;
    uint64_t total = 0;

    /* pass 1: per-chunk byte totals -> chunk base offsets */
    chunk_bytes = (uint64_t*)__builtin_malloc((nchunks + 1) * sizeof(uint64_t));
    if (!chunk_bytes) return UINT64_MAX;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < (int64_t)nchunks; c++) {
        uint64_t b = (uint64_t)c * CHUNK, e = b + CHUNK < nrows ? b + CHUNK : nrows, t = 0;
        char buf[64];
        for (uint64_t i = b; i < e; i++) t += (uint64_t)gen_value(s, row0 + i, buf);
        chunk_bytes[c] = t;
    }
    for (uint64_t c = 0; c < nchunks; c++) {
        uint64_t t = chunk_bytes[c];
        chunk_bytes[c] = total;
        total += t;
    }
    if (o32 && total > 0xFFFFFFFFull) {
        __builtin_free(chunk_bytes);
        return UINT64_MAX;
    }
    /* pass 2: fill */
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < (int64_t)nchunks; c++) {
        uint64_t b = (uint64_t)c * CHUNK, e = b + CHUNK < nrows ? b + CHUNK : nrows;
        uint64_t pos = chunk_bytes[c];
        char buf[64];
        for (uint64_t i = b; i < e; i++) {
            int n = gen_value(s, row0 + i, buf);
            if (o32) o32[i] = (uint32_t)pos; else o64[i] = pos;
            memcpy(data + pos, buf, (size_t)n);
            pos += (uint64_t)n;
        }
    }
    if (o32) o32[nrows] = (uint32_t)total; else o64[nrows] = total;
    __builtin_free(chunk_bytes);
    return total;
}

/* The numeric value behind row `row` (ground truth for tests). */
DG_API uint64_t dg_value_u64(const dg_spec* s, uint64_t row) {
    switch (s->kind) {
    case DG_SEQ_PERM: case DG_NAME: case DG_SURNAME: case DG_PRODUCT: case DG_PRICE:
        return feistel_perm(row, s->domain, s->seed);
    case DG_SEQ: return row;
    case DG_UNIFORM: return s->base + bounded(rnd(s->seed, row, 1), s->domain);
    case DG_UNIFORM_PERM:
        return s->base + feistel_perm(bounded(rnd(s->seed, row, 1), s->domain), s->domain, s->seed ^ 0x51ED);
    case DG_FK_SUBSET: return feistel_perm(bounded(rnd(s->seed ^ 0xF00Dull, row, 1), s->base), s->domain, s->seed);
    case DG_RANDKEY: return feistel_perm(s->base ? bounded(rnd(s->seed ^ 0xF00Dull, row, 1), s->base) : row, (uint64_t)1 << 62, s->seed);
    default: return 0;
    }
}

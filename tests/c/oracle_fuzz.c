/* oracle_fuzz.c — the CPU checker under AddressSanitizer + UBSan (`make asan`, tests/test_asan.py).
 *
 * The oracle is what every parity claim is measured against, so its own memory safety is checked the hard way:
 * this driver is compiled TOGETHER with oracle/csvplus_oracle.c with -fsanitize=address,undefined and feeds it
 *   - random, mostly malformed CSV texts (the shapes of tests/test_csv_ingest.py: unbalanced quotes, bare CR,
 *     blank lines, comments) through the two-pass orc_csv_parse exactly as oracle/orc.py sizes its buffers,
 *     then writes the parsed columns back with orc_csv_write;
 *   - random key tables (empty values, NUL / 0xFF bytes, heavy duplicates, keys up to 300 bytes) through
 *     orc_index_build (both sort modes), orc_first_dup, orc_find, orc_has and orc_join (full and prefix).
 * Buffers are malloc'ed at their exact sizes so that one byte out of bounds aborts the run.  Exit status 0 and
 * the line "ORACLE_FUZZ_OK <digest>" mean no finding; the digest pins the run (deterministic PRNG). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/csvplus_oracle.c"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) {
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
    return g_state;
}
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

typedef struct { uint8_t* data; uint64_t* offs; uint64_t n; } table_col;

static table_col make_col(uint64_t n, uint32_t maxlen, const char* alphabet, uint32_t na, uint32_t distinct) {
    table_col c;
    uint8_t** pool = (uint8_t**)malloc(sizeof(uint8_t*) * (distinct ? distinct : 1));
    uint32_t* plen = (uint32_t*)malloc(sizeof(uint32_t) * (distinct ? distinct : 1));
    for (uint32_t i = 0; i < distinct; i++) {
        plen[i] = below(maxlen + 1);
        pool[i] = (uint8_t*)malloc(plen[i] ? plen[i] : 1);
        for (uint32_t k = 0; k < plen[i]; k++) pool[i][k] = (uint8_t)alphabet[below(na)];
    }
    c.n = n;
    c.offs = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
    uint64_t total = 0;
    uint32_t* pick = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint64_t r = 0; r < n; r++) { pick[r] = below(distinct); c.offs[r] = total; total += plen[pick[r]]; }
    c.offs[n] = total;
    c.data = (uint8_t*)malloc(total ? total : 1);
    for (uint64_t r = 0; r < n; r++) memcpy(c.data + c.offs[r], pool[pick[r]], plen[pick[r]]);
    for (uint32_t i = 0; i < distinct; i++) free(pool[i]);
    free(pool); free(plen); free(pick);
    return c;
}
static orc_strcol as_strcol(const table_col* c) {
    orc_strcol s;
    s.data = c->data; s.offsets = c->offs; s.nrows = c->n; s.offset_bits = 64; s.mem = 0;
    return s;
}
static void free_col(table_col* c) { free(c->data); free(c->offs); }

static uint64_t fuzz_csv(int rounds) {
    static const char alpha[] = "ab ,\"\n\rz#";
    uint64_t digest = 1469598103934665603ull;
    for (int it = 0; it < rounds; it++) {
        const uint64_t size = below(it % 7 == 0 ? 4000 : 200);
        uint8_t* text = (uint8_t*)malloc(size ? size : 1);
        for (uint64_t i = 0; i < size; i++) text[i] = (uint8_t)alpha[below(it & 1 ? 9 : 8)];
        for (int variant = 0; variant < 4; variant++) {
            orc_csv_opts o;
            memset(&o, 0, sizeof o);
            o.comma = ',';
            o.comment = variant == 3 ? '#' : 0;
            o.trim_leading_space = variant == 2;
            o.fields_per_record = variant == 1 ? 0 : -1;
            o.skip_records = below(3);
            const int32_t ncols = 1 + (int32_t)below(3);
            int32_t col_index[3];
            for (int32_t c = 0; c < ncols; c++) col_index[c] = (int32_t)below(4);
            uint64_t nbytes[3] = {0, 0, 0};
            int32_t ek = 0; uint64_t er = 0;
            const uint64_t n = orc_csv_parse(text, size, &o, col_index, ncols, nbytes, NULL, NULL, &ek, &er);
            uint8_t* datas[3]; uint64_t* offs[3];
            for (int32_t c = 0; c < ncols; c++) {   /* oracle/orc.py: room for the record that fails and is rolled back */
                datas[c] = (uint8_t*)malloc(nbytes[c] + size + 1);
                offs[c] = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
            }
            int32_t ek2 = 0; uint64_t er2 = 0;
            uint64_t nb2[3];
            const uint64_t n2 = orc_csv_parse(text, size, &o, col_index, ncols, nb2, datas, offs, &ek2, &er2);
            if (n2 != n || ek2 != ek || er2 != er) { printf("two passes disagree\n"); exit(2); }
            orc_strcol cols[3];
            for (int32_t c = 0; c < ncols; c++) {
                if (nb2[c] != nbytes[c] || offs[c][n] != nbytes[c]) { printf("sizes disagree\n"); exit(2); }
                cols[c].data = datas[c]; cols[c].offsets = offs[c]; cols[c].nrows = n; cols[c].offset_bits = 64; cols[c].mem = 0;
                digest = orc_fnv1a64(datas[c], nbytes[c], digest);
            }
            const uint64_t wsize = orc_csv_write(cols, ncols, NULL, NULL, 0);
            uint8_t* w = (uint8_t*)malloc(wsize ? wsize : 1);
            if (orc_csv_write(cols, ncols, NULL, w, wsize) != wsize) { printf("csv_write sizes disagree\n"); exit(2); }
            digest = orc_fnv1a64(w, wsize, digest);
            digest = orc_fnv1a64(&ek, sizeof ek, digest);
            free(w);
            for (int32_t c = 0; c < ncols; c++) { free(datas[c]); free(offs[c]); }
        }
        free(text);
    }
    return digest;
}

static uint64_t fuzz_index(int rounds) {
    static const char alpha[] = {'a', 'b', 0, (char)0xFF, 'z'};
    uint64_t digest = 1469598103934665603ull;
    for (int it = 0; it < rounds; it++) {
        const int32_t ncols = 1 + (int32_t)below(3);
        const uint64_t n = below(it % 5 == 0 ? 3000 : 200), m = below(400);
        const uint32_t maxlen = it % 4 == 0 ? 300 : 6;
        table_col b[3], p[3];
        orc_strcol bc[3], pc[3];
        for (int32_t c = 0; c < ncols; c++) {
            b[c] = make_col(n, maxlen, alpha, 5, 1 + below(n ? (uint32_t)n : 1));
            p[c] = make_col(m, maxlen, alpha, 5, 1 + below(50));
            bc[c] = as_strcol(&b[c]);
            pc[c] = as_strcol(&p[c]);
        }
        uint32_t* perm = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
        uint32_t* perm2 = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
        orc_index_build(bc, ncols, ORC_SORT_STABLE, perm);
        orc_index_build(bc, ncols, ORC_SORT_GO_PDQSORT, perm2);
        digest = orc_fnv1a64(perm, sizeof(uint32_t) * n, digest);
        const uint64_t fd = orc_first_dup(bc, ncols, perm);
        digest = orc_fnv1a64(&fd, sizeof fd, digest);
        for (int32_t k = 1; k <= ncols; k++) {   /* full key and every prefix */
            uint32_t* lo = (uint32_t*)malloc(sizeof(uint32_t) * (m ? m : 1));
            uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * (m ? m : 1));
            const uint64_t total = n ? orc_join(bc, ncols, perm, pc, k, NULL, 0, 7, lo, cnt, NULL, NULL, 0) : 0;
            uint64_t* pi = (uint64_t*)malloc(sizeof(uint64_t) * (total ? total : 1));
            uint32_t* br = (uint32_t*)malloc(sizeof(uint32_t) * (total ? total : 1));
            if (n && orc_join(bc, ncols, perm, pc, k, NULL, 0, 7, NULL, NULL, pi, br, total) != total) { printf("join totals disagree\n"); exit(2); }
            digest = orc_fnv1a64(br, sizeof(uint32_t) * total, digest);
            if (n && m) {
                orc_strval v[3];
                for (int32_t c = 0; c < k; c++) { v[c].data = p[c].data + p[c].offs[0]; v[c].len = p[c].offs[1] - p[c].offs[0]; }
                uint64_t flo = 0, fhi = 0;
                orc_find(bc, perm, v, k, &flo, &fhi);
                const int32_t h = orc_has(bc, perm, v, k);
                if ((fhi > flo) != (h != 0) && k == ncols) { printf("find / has disagree\n"); exit(2); }
                digest = orc_fnv1a64(&flo, sizeof flo, digest);
            }
            free(lo); free(cnt); free(pi); free(br);
        }
        free(perm); free(perm2);
        for (int32_t c = 0; c < ncols; c++) { free_col(&b[c]); free_col(&p[c]); }
    }
    return digest;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    const uint64_t d1 = fuzz_csv(rounds);
    const uint64_t d2 = fuzz_index(rounds / 3 + 1);
    printf("ORACLE_FUZZ_OK %016llx %016llx\n", (unsigned long long)d1, (unsigned long long)d2);
    return 0;
}

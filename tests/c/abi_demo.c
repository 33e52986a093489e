/* abi_demo.c — the C ABI used from plain C (what cgo compiles against): TestIndexImpl's rows
 * (csvplus_test.go:198-246) indexed on (x,y,z), found, joined and written as CSV.  Prints one line per check;
 * exit status 0 = all as expected, 2 = no GPU (the library has no CPU path), 1 = wrong result. */
#include <stdio.h>
#include <string.h>

#include "csvplus_hip.h"

static const char* X[] = {"1", "5", "0", "8", "7", "5", "2"};
static const char* Y[] = {"2", "6", "5", "9", "4", "6", "6"};
static const char* Z[] = {"3", "8", "3", "1", "0", "9", "7"};
static const char* JUNK[] = {"zzz", "nnn", "xxx", "aaa", "bbb", "iii", "mmm"};
enum { N = 7 };

static cph_strcol column(const char** v, uint8_t* data, uint32_t* offs) {
    uint32_t pos = 0;
    for (int i = 0; i < N; i++) {
        offs[i] = pos;
        memcpy(data + pos, v[i], strlen(v[i]));
        pos += (uint32_t)strlen(v[i]);
    }
    offs[N] = pos;
    cph_strcol c;
    memset(&c, 0, sizeof c);
    c.data = data;
    c.offsets = offs;
    c.nrows = N;
    c.offset_bits = 32;
    c.mem = CPH_MEM_HOST;
    return c;
}

int main(void) {
    cph_ctx* ctx = NULL;
    int32_t rc = cph_ctx_create(0, &ctx);
    if (rc == CPH_ERR_NO_DEVICE) {
        printf("no usable GPU: libcsvplus_hip has no CPU fallback\n");
        return 2;
    }
    if (rc != CPH_OK) return 1;
    uint8_t d[4][64];
    uint32_t o[4][N + 1];
    cph_strcol key[3] = {column(X, d[0], o[0]), column(Y, d[1], o[1]), column(Z, d[2], o[2])};
    cph_strcol junk = column(JUNK, d[3], o[3]);
    int bad = 0;

    cph_index* ix = NULL;
    uint64_t dup = 0;
    rc = cph_index_build(ctx, key, 3, 0, &ix, &dup);
    if (rc != CPH_OK) { printf("index_build: %s\n", cph_last_error(ctx)); return 1; }
    const uint32_t* perm = NULL;
    uint64_t n = 0;
    cph_index_perm(ix, CPH_MEM_HOST, &perm, &n);
    static const char* want_order[] = {"xxx", "zzz", "mmm", "nnn", "iii", "bbb", "aaa"};
    for (uint64_t i = 0; i < n; i++) bad |= strcmp(JUNK[perm[i]], want_order[i]) != 0;
    printf("sorted order %s\n", bad ? "WRONG" : "ok");

    cph_strval v56[2] = {{(const uint8_t*)"5", 1}, {(const uint8_t*)"6", 1}};
    uint64_t lo = 0, hi = 0;
    rc = cph_index_find(ctx, ix, v56, 2, &lo, &hi);
    bad |= rc != CPH_OK || lo != 3 || hi != 5;
    printf("find(5,6) = [%llu,%llu) %s\n", (unsigned long long)lo, (unsigned long long)hi, (lo == 3 && hi == 5) ? "ok" : "WRONG");

    /* join the table with itself on x only (prefix join): x=5 occurs twice -> 1+2+1+1+1+2+1 = 9 pairs */
    cph_matches* m = NULL;
    rc = cph_join_probe(ctx, ix, key, 1, NULL, 32, 0, 0, 0, 1, CPH_MEM_HOST, &m);
    bad |= rc != CPH_OK || m->nmatches != 9;
    printf("prefix self-join: %llu pairs %s\n", (unsigned long long)(m ? m->nmatches : 0), (m && m->nmatches == 9) ? "ok" : "WRONG");

    /* Join(...).ToCsv: stream column x + the index side's junk through the pair list */
    if (m && m->nmatches == 9) {
        cph_strcol cols[2] = {key[0], junk};
        cph_rowsel sel[2];
        memset(sel, 0, sizeof sel);
        sel[0].ids = m->probe_idx; sel[0].bits = 64;
        sel[1].ids = m->build_row; sel[1].bits = 32;
        cph_strval header[2] = {{(const uint8_t*)"x", 1}, {(const uint8_t*)"junk", 4}};
        cph_bytes* text = NULL;
        rc = cph_csv_write_rows(ctx, cols, sel, 2, m->nmatches, header, CPH_MEM_HOST, &text);
        static const char want[] = "x,junk\n1,zzz\n5,nnn\n5,iii\n0,xxx\n8,aaa\n7,bbb\n5,nnn\n5,iii\n2,mmm\n";
        const int ok = rc == CPH_OK && text->size == sizeof want - 1 && memcmp(text->data, want, sizeof want - 1) == 0;
        bad |= !ok;
        printf("joined csv %s\n", ok ? "ok" : "WRONG");
        cph_bytes_release(text);
    }
    cph_matches_release(m);
    cph_index_destroy(ix);
    cph_ctx_destroy(ctx);
    return bad ? 1 : 0;
}

/*
 * csvplus_oracle.c — CPU restatement of csvplus's Index-build + Join path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under csvplus_amd/ (the product) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the reported
 * CPU baseline.
 *
 * Every function cites the reference lines it restates (maxim2266/csvplus,
 * file csvplus.go unless stated otherwise).  The reference is Go and there is
 * no Go toolchain in this image, so the reference itself cannot be executed
 * here.  Pinning status: the oracle is checked against the reference's only
 * literal known-answer vector (TestIndexImpl, csvplus_test.go:198-246) and
 * against re-statements of its structural tests on fixtures of the same shape
 * (TestSorted :454-514, TestSimpleUniqueJoin :368-452, TestMultiIndex
 * :573-649, TestExcept :651-693, TestErrors :826-841) — see
 * tests/test_oracle.py.  The intra-group order of rows with EQUAL keys is not
 * pinned by any reference test (sort.Sort is unstable); the
 * canonical order here is the stable one (input order), and
 * ORC_SORT_GO_PDQSORT offers an emulation of Go's pdqsort written from memory
 * of go1.19+ src/sort/zsortinterface.go, which is UNVERIFIED (parity class P2).
 *
 * Rows are represented SoA: one Arrow-style string column per key column.
 * A Go `Row` (map[string]string, :59) lookup `row[col]` becomes
 * value(col, rowid).  createIndex guarantees all key columns exist (:723-727),
 * so "missing" never occurs on this path.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    const uint8_t* data;
    const void*    offsets;
    uint64_t       nrows;
    int32_t        offset_bits;
    int32_t        mem; /* layout-compatible with cph_strcol; ignored (host only) */
} orc_strcol;

typedef struct {
    const uint8_t* data;
    uint64_t       len;
} orc_strval;

enum { ORC_SORT_STABLE = 0, ORC_SORT_GO_PDQSORT = 1 };

static inline uint64_t col_off(const orc_strcol* c, uint64_t i) {
    return c->offset_bits == 32 ? ((const uint32_t*)c->offsets)[i] : ((const uint64_t*)c->offsets)[i];
}

static inline void col_value(const orc_strcol* c, uint64_t row, const uint8_t** p, uint64_t* len) {
    uint64_t b = col_off(c, row), e = col_off(c, row + 1);
    *p = c->data + b;
    *len = e - b;
}

/* strings.Compare (Go stdlib; call sites :798, :911): unsigned bytewise
 * lexicographic; if one is a prefix of the other the shorter is smaller. */
static inline int go_strings_compare(const uint8_t* a, uint64_t alen, const uint8_t* b, uint64_t blen) {
    uint64_t n = alen < blen ? alen : blen;
    int c = n ? memcmp(a, b, (size_t)n) : 0;
    if (c) return c < 0 ? -1 : 1;
    return alen < blen ? -1 : (alen > blen ? 1 : 0);
}

/* ---- the sort.Interface of indexImpl (:791-807) ---------------------------- */

typedef struct {
    const orc_strcol* cols;
    int32_t           ncols;
    uint32_t*         rows; /* index.rows: here row ids */
    uint64_t          less_calls;
} orc_index_impl;

/* indexImpl.Less (:794-807): column by column strings.Compare, first
 * non-equal column decides; all equal -> false. */
static inline int impl_less_ids(const orc_index_impl* ix, uint32_t left, uint32_t right) {
    for (int32_t c = 0; c < ix->ncols; c++) {
        const uint8_t *a, *b;
        uint64_t al, bl;
        col_value(&ix->cols[c], left, &a, &al);
        col_value(&ix->cols[c], right, &b, &bl);
        switch (go_strings_compare(a, al, b, bl)) {
        case -1: return 1;
        case 1: return 0;
        }
    }
    return 0;
}
static inline int impl_less(orc_index_impl* ix, int64_t i, int64_t j) {
    ix->less_calls++;
    return impl_less_ids(ix, ix->rows[i], ix->rows[j]);
}
/* indexImpl.Swap (:792) */
static inline void impl_swap(orc_index_impl* ix, int64_t i, int64_t j) {
    uint32_t t = ix->rows[i];
    ix->rows[i] = ix->rows[j];
    ix->rows[j] = t;
}

/* ---- canonical stable sort: merge sort on row ids --------------------------- */

static void merge_sort(orc_index_impl* ix, uint32_t* a, uint32_t* tmp, int64_t n) {
    if (n <= 12) { /* insertion sort, stable */
        for (int64_t i = 1; i < n; i++) {
            uint32_t v = a[i];
            int64_t j = i;
            while (j > 0 && impl_less_ids(ix, v, a[j - 1])) { a[j] = a[j - 1]; j--; }
            a[j] = v;
        }
        return;
    }
    int64_t h = n / 2;
    merge_sort(ix, a, tmp, h);
    merge_sort(ix, a + h, tmp, n - h);
    if (!impl_less_ids(ix, a[h], a[h - 1])) return; /* already ordered */
    memcpy(tmp, a, (size_t)h * sizeof(uint32_t));
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take from the right run only when strictly less: stability */
        if (impl_less_ids(ix, a[j], tmp[i])) a[k++] = a[j++]; else a[k++] = tmp[i++];
    }
    while (i < h) a[k++] = tmp[i++];
}

/* ---- Go's sort.Sort = pdqsort (go1.19+), restated from memory: UNVERIFIED ----
 * Call site :736 `sort.Sort(&index.impl)`.  Kept bug-for-bug as remembered
 * (e.g. partialInsertionSort's `j >= 1` bound). */

enum { HINT_UNKNOWN = 0, HINT_INCREASING = 1, HINT_DECREASING = 2 };

static int bits_len(uint64_t x) { int n = 0; while (x) { n++; x >>= 1; } return n; }

static void pdq_insertion_sort(orc_index_impl* d, int64_t a, int64_t b) {
    for (int64_t i = a + 1; i < b; i++)
        for (int64_t j = i; j > a && impl_less(d, j, j - 1); j--) impl_swap(d, j, j - 1);
}
static void pdq_sift_down(orc_index_impl* d, int64_t lo, int64_t hi, int64_t first) {
    int64_t root = lo;
    for (;;) {
        int64_t child = 2 * root + 1;
        if (child >= hi) return;
        if (child + 1 < hi && impl_less(d, first + child, first + child + 1)) child++;
        if (!impl_less(d, first + root, first + child)) return;
        impl_swap(d, first + root, first + child);
        root = child;
    }
}
static void pdq_heap_sort(orc_index_impl* d, int64_t a, int64_t b) {
    int64_t first = a, lo = 0, hi = b - a;
    for (int64_t i = (hi - 1) / 2; i >= 0; i--) pdq_sift_down(d, i, hi, first);
    for (int64_t i = hi - 1; i >= 0; i--) {
        impl_swap(d, first, first + i);
        pdq_sift_down(d, lo, i, first);
    }
}
static void pdq_break_patterns(orc_index_impl* d, int64_t a, int64_t b) {
    int64_t length = b - a;
    if (length >= 8) {
        uint64_t random = (uint64_t)length;
        uint64_t modulus = (uint64_t)1 << bits_len((uint64_t)length);
        int64_t idx = a + (length / 4) * 2 - 1;
        for (int i = 0; i < 3; i++) {
            random ^= random << 13;
            random ^= random >> 7;
            random ^= random << 17;
            int64_t other = (int64_t)(random & (modulus - 1));
            if (other >= length) other -= length;
            impl_swap(d, idx - 1 + i, a + other);
        }
    }
}
static void pdq_order2(orc_index_impl* d, int64_t* a, int64_t* b, int* swaps) {
    if (impl_less(d, *b, *a)) {
        (*swaps)++;
        int64_t t = *a; *a = *b; *b = t;
    }
}
static int64_t pdq_median(orc_index_impl* d, int64_t a, int64_t b, int64_t c, int* swaps) {
    pdq_order2(d, &a, &b, swaps);
    pdq_order2(d, &b, &c, swaps);
    pdq_order2(d, &a, &b, swaps);
    return b;
}
static int64_t pdq_choose_pivot(orc_index_impl* d, int64_t a, int64_t b, int* hint) {
    int64_t l = b - a;
    int swaps = 0;
    int64_t i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) {
        if (l >= 50) {
            i = pdq_median(d, i - 1, i, i + 1, &swaps);
            j = pdq_median(d, j - 1, j, j + 1, &swaps);
            k = pdq_median(d, k - 1, k, k + 1, &swaps);
        }
        j = pdq_median(d, i, j, k, &swaps);
    }
    *hint = swaps == 0 ? HINT_INCREASING : (swaps == 12 ? HINT_DECREASING : HINT_UNKNOWN);
    return j;
}
static void pdq_reverse_range(orc_index_impl* d, int64_t a, int64_t b) {
    int64_t i = a, j = b - 1;
    while (i < j) { impl_swap(d, i, j); i++; j--; }
}
static int pdq_partial_insertion_sort(orc_index_impl* d, int64_t a, int64_t b) {
    int64_t i = a + 1;
    for (int step = 0; step < 5; step++) {
        while (i < b && !impl_less(d, i, i - 1)) i++;
        if (i == b) return 1;
        if (b - a < 50) return 0;
        impl_swap(d, i, i - 1);
        if (i - a >= 2)
            for (int64_t j = i - 1; j >= 1; j--) {
                if (!impl_less(d, j, j - 1)) break;
                impl_swap(d, j, j - 1);
            }
        if (b - i >= 2)
            for (int64_t j = i + 1; j < b; j++) {
                if (!impl_less(d, j, j - 1)) break;
                impl_swap(d, j, j - 1);
            }
    }
    return 0;
}
static int64_t pdq_partition_equal(orc_index_impl* d, int64_t a, int64_t b, int64_t pivot) {
    impl_swap(d, a, pivot);
    int64_t i = a + 1, j = b - 1;
    for (;;) {
        while (i <= j && !impl_less(d, a, i)) i++;
        while (i <= j && impl_less(d, a, j)) j--;
        if (i > j) break;
        impl_swap(d, i, j);
        i++; j--;
    }
    return i;
}
static int64_t pdq_partition(orc_index_impl* d, int64_t a, int64_t b, int64_t pivot, int* already) {
    impl_swap(d, a, pivot);
    int64_t i = a + 1, j = b - 1;
    while (i <= j && impl_less(d, i, a)) i++;
    while (i <= j && !impl_less(d, j, a)) j--;
    if (i > j) {
        impl_swap(d, j, a);
        *already = 1;
        return j;
    }
    impl_swap(d, i, j);
    i++; j--;
    for (;;) {
        while (i <= j && impl_less(d, i, a)) i++;
        while (i <= j && !impl_less(d, j, a)) j--;
        if (i > j) break;
        impl_swap(d, i, j);
        i++; j--;
    }
    impl_swap(d, j, a);
    *already = 0;
    return j;
}
static void pdqsort(orc_index_impl* d, int64_t a, int64_t b, int limit) {
    int was_balanced = 1, was_partitioned = 1;
    for (;;) {
        int64_t length = b - a;
        if (length <= 12) { pdq_insertion_sort(d, a, b); return; }
        if (limit == 0) { pdq_heap_sort(d, a, b); return; }
        if (!was_balanced) { pdq_break_patterns(d, a, b); limit--; }
        int hint;
        int64_t pivot = pdq_choose_pivot(d, a, b, &hint);
        if (hint == HINT_DECREASING) {
            pdq_reverse_range(d, a, b);
            pivot = (b - 1) - (pivot - a);
            hint = HINT_INCREASING;
        }
        if (was_balanced && was_partitioned && hint == HINT_INCREASING)
            if (pdq_partial_insertion_sort(d, a, b)) return;
        if (a > 0 && !impl_less(d, a - 1, pivot)) {
            a = pdq_partition_equal(d, a, b, pivot);
            continue;
        }
        int already;
        int64_t mid = pdq_partition(d, a, b, pivot, &already);
        was_partitioned = already;
        int64_t left_len = mid - a, right_len = b - mid;
        int64_t balance_threshold = length / 8;
        if (left_len < right_len) {
            was_balanced = left_len >= balance_threshold;
            pdqsort(d, a, mid, limit);
            a = mid + 1;
        } else {
            was_balanced = right_len >= balance_threshold;
            pdqsort(d, mid + 1, b, limit);
            b = mid;
        }
    }
}

/* ---- createIndex (:707-738): drain rows, sort.Sort ---------------------------
 * perm[i] = original row id of the row at sorted position i.  Returns the
 * number of Less calls (cost-model statistic), or UINT64_MAX on error. */
ORC_API uint64_t orc_index_build(const orc_strcol* cols, int32_t ncols, int32_t mode, uint32_t* perm) {
    if (!cols || ncols <= 0 || !perm) return UINT64_MAX;
    const uint64_t n = cols[0].nrows;
    if (n > 0xFFFFFFFFull) return UINT64_MAX;
    orc_index_impl ix = {cols, ncols, perm, 0};
    for (uint64_t i = 0; i < n; i++) perm[i] = (uint32_t)i; /* :729 append in stream order */
    if (n <= 1) return 0; /* sort.Sort: n <= 1 returns */
    if (mode == ORC_SORT_GO_PDQSORT) {
        pdqsort(&ix, 0, (int64_t)n, bits_len(n));
    } else {
        uint32_t* tmp = (uint32_t*)malloc((size_t)(n / 2 + 1) * sizeof(uint32_t));
        if (!tmp) return UINT64_MAX;
        merge_sort(&ix, perm, tmp, (int64_t)n);
        free(tmp);
    }
    return ix.less_calls;
}

/* equalRows (:759-767) on two row ids */
static inline int equal_rows(const orc_strcol* cols, int32_t ncols, uint32_t r1, uint32_t r2) {
    for (int32_t c = 0; c < ncols; c++) {
        const uint8_t *a, *b;
        uint64_t al, bl;
        col_value(&cols[c], r1, &a, &al);
        col_value(&cols[c], r2, &b, &bl);
        if (al != bl || (al && memcmp(a, b, (size_t)al))) return 0;
    }
    return 1;
}

/* createUniqueIndex adjacent scan (:742-753): first i>=1 with
 * rows[i-1]==rows[i]; UINT64_MAX when none (or fewer than 2 rows). */
ORC_API uint64_t orc_first_dup(const orc_strcol* cols, int32_t ncols, const uint32_t* perm) {
    const uint64_t n = cols[0].nrows;
    for (uint64_t i = 1; i < n; i++)
        if (equal_rows(cols, ncols, perm[i - 1], perm[i])) return i;
    return UINT64_MAX;
}

/* ---- indexImpl.cmp / first / has / find (:893-920, :870-891) ----------------- */

/* cmp (:907-920): row i's first nvalues key columns vs values:
 * 1 if greater, 0 if less, eq if equal. */
static inline int impl_cmp(const orc_strcol* cols, const uint32_t* perm, uint64_t i, const orc_strval* values,
                           int32_t nvalues, int eq) {
    const uint32_t row = perm[i];
    for (int32_t j = 0; j < nvalues; j++) {
        const uint8_t* a;
        uint64_t al;
        col_value(&cols[j], row, &a, &al);
        switch (go_strings_compare(a, al, values[j].data, values[j].len)) {
        case 1: return 1;
        case -1: return 0;
        }
    }
    return eq;
}

/* sort.Search (Go stdlib; call sites :831, :881, :885, :894):
 * smallest i in [0,n) with f(i) true, else n. */
#define GO_SORT_SEARCH(result, n_, pred)          \
    do {                                          \
        uint64_t i_ = 0, j_ = (n_);               \
        while (i_ < j_) {                         \
            uint64_t h = (i_ + j_) >> 1;          \
            if (!(pred)) i_ = h + 1; else j_ = h; \
        }                                         \
        (result) = i_;                            \
    } while (0)

/* first (:893-897) */
static inline uint64_t impl_first(const orc_strcol* cols, const uint32_t* perm, uint64_t n, const orc_strval* values,
                                  int32_t nvalues) {
    uint64_t r;
    GO_SORT_SEARCH(r, n, impl_cmp(cols, perm, h, values, nvalues, 1));
    return r;
}

/* find (:870-891): [lower, upper) */
ORC_API void orc_find(const orc_strcol* cols, const uint32_t* perm, const orc_strval* values, int32_t nvalues,
                      uint64_t* lower, uint64_t* upper) {
    const uint64_t n = cols[0].nrows;
    if (nvalues == 0) { *lower = 0; *upper = n; return; } /* :872-874 */
    uint64_t up, lo;
    GO_SORT_SEARCH(up, n, impl_cmp(cols, perm, h, values, nvalues, 0));  /* :881 */
    GO_SORT_SEARCH(lo, up, impl_cmp(cols, perm, h, values, nvalues, 1)); /* :885 */
    *lower = lo;
    *upper = up;
}

/* has (:899-905) */
ORC_API int32_t orc_has(const orc_strcol* cols, const uint32_t* perm, const orc_strval* values, int32_t nvalues) {
    const uint64_t n = cols[0].nrows;
    uint64_t i = impl_first(cols, perm, n, values, nvalues);
    return i < n && !impl_cmp(cols, perm, i, values, nvalues, 0);
}

/* ---- Join (:545-569) ------------------------------------------------------------
 * For each probe row (stream order): values = SelectValues(columns) (:556),
 * i = first(values) (:559), emit while i<n && !cmp(i,values,false).
 *
 * lo/cnt (nprobe entries each, may be NULL) receive first() and the number of
 * loop iterations.  If probe_idx/build_row are non-NULL they receive up to
 * cap pairs in emission order.  Returns the total number of matches.
 * row_sel (optional) selects probe rows; probe_base offsets probe_idx.
 */
ORC_API uint64_t orc_join(const orc_strcol* bcols, int32_t nbcols, const uint32_t* perm, const orc_strcol* pcols,
                          int32_t npcols, const uint32_t* row_sel, uint64_t nsel, uint64_t probe_base, uint32_t* lo,
                          uint32_t* cnt, uint64_t* probe_idx, uint32_t* build_row, uint64_t cap) {
    (void)nbcols;
    const uint64_t n = bcols[0].nrows;
    const uint64_t nprobe = row_sel ? nsel : pcols[0].nrows;
    uint64_t total = 0;
    orc_strval values[64];
    for (uint64_t p = 0; p < nprobe; p++) {
        const uint64_t prow = row_sel ? row_sel[p] : p;
        for (int32_t c = 0; c < npcols; c++) col_value(&pcols[c], prow, &values[c].data, &values[c].len);
        uint64_t first = impl_first(bcols, perm, n, values, npcols);
        uint64_t i = first;
        for (; i < n && !impl_cmp(bcols, perm, i, values, npcols, 0); i++) {
            if (probe_idx && total < cap) {
                probe_idx[total] = probe_base + p;
                build_row[total] = perm[i];
            }
            total++;
        }
        if (lo) lo[p] = (uint32_t)first;
        if (cnt) cnt[p] = (uint32_t)(i - first);
    }
    return total;
}

/* ---- digest used by parity tests (FNV-1a 64 over raw bytes) ------------------- */
ORC_API uint64_t orc_fnv1a64(const void* p, uint64_t nbytes, uint64_t h) {
    const uint8_t* b = (const uint8_t*)p;
    if (h == 0) h = 0xCBF29CE484222325ull;
    for (uint64_t i = 0; i < nbytes; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

/* ---- ToCsv (:379-406) through Go's encoding/csv Writer, default settings ---------------------
 * The Writer lives in the Go standard library (not under /root/reference); restated from its
 * documented behaviour and source as remembered (go1.19+ encoding/csv/writer.go):
 *   fieldNeedsQuotes(field): "" -> false; `\.` -> true; contains Comma, '"', '\r' or '\n' -> true;
 *                            otherwise unicode.IsSpace(first rune).
 *   quoted field: '"' + field with every '"' doubled + '"'   (UseCRLF = false: '\r', '\n' verbatim)
 *   record: fields joined by Comma (','), terminated by '\n'.
 * Parity of this restatement is additionally cross-checked against Python's csv module on the
 * subset where both agree (tests/test_oracle.py).                                              */
static int go_unicode_is_space_first_rune(const uint8_t* p, uint64_t len) {
    if (!len) return 0;
    uint32_t b0 = p[0], b1 = len > 1 ? p[1] : 0, b2 = len > 2 ? p[2] : 0;
    if (b0 < 0x80) return b0 == ' ' || (b0 >= 9 && b0 <= 13);
    if (b0 == 0xC2 && len >= 2) return b1 == 0x85 || b1 == 0xA0;
    if (len < 3) return 0;
    if (b0 == 0xE1) return b1 == 0x9A && b2 == 0x80;
    if (b0 == 0xE2) {
        if (b1 == 0x80) return (b2 >= 0x80 && b2 <= 0x8A) || b2 == 0xA8 || b2 == 0xA9 || b2 == 0xAF;
        return b1 == 0x81 && b2 == 0x9F;
    }
    return b0 == 0xE3 && b1 == 0x80 && b2 == 0x80;
}
static int go_csv_field_needs_quotes(const uint8_t* p, uint64_t len) {
    if (len == 0) return 0;
    if (len == 2 && p[0] == '\\' && p[1] == '.') return 1;
    for (uint64_t i = 0; i < len; i++)
        if (p[i] == ',' || p[i] == '"' || p[i] == '\r' || p[i] == '\n') return 1;
    return go_unicode_is_space_first_rune(p, len);
}
static uint64_t go_csv_put_field(uint8_t* out, uint64_t pos, uint64_t cap, const uint8_t* p, uint64_t len) {
#define PUT(b) do { if (out && pos < cap) out[pos] = (uint8_t)(b); pos++; } while (0)
    if (!go_csv_field_needs_quotes(p, len)) {
        for (uint64_t i = 0; i < len; i++) PUT(p[i]);
        return pos;
    }
    PUT('"');
    for (uint64_t i = 0; i < len; i++) {
        if (p[i] == '"') PUT('"');
        PUT(p[i]);
    }
    PUT('"');
    return pos;
#undef PUT
}
/* Writes header (if nheader > 0) + all rows; returns the total size (call with out=NULL to size). */
ORC_API uint64_t orc_csv_write(const orc_strcol* cols, int32_t ncols, const orc_strval* header, uint8_t* out, uint64_t cap) {
    uint64_t pos = 0;
    if (header) {
        for (int32_t c = 0; c < ncols; c++) {
            if (c) { if (out && pos < cap) out[pos] = ','; pos++; }
            pos = go_csv_put_field(out, pos, cap, header[c].data, header[c].len);
        }
        if (out && pos < cap) out[pos] = '\n';
        pos++;
    }
    const uint64_t n = cols[0].nrows;
    for (uint64_t r = 0; r < n; r++) {
        for (int32_t c = 0; c < ncols; c++) {
            const uint8_t* p;
            uint64_t len;
            col_value(&cols[c], r, &p, &len);
            if (c) { if (out && pos < cap) out[pos] = ','; pos++; }
            pos = go_csv_put_field(out, pos, cap, p, len);
        }
        if (out && pos < cap) out[pos] = '\n';
        pos++;
    }
    return pos;
}

/* ---- CSV ingest: Go's encoding/csv Reader as configured by csvplus's Reader.Iterate
 * (csvplus.go:1080-1146: Comma, Comment, TrimLeadingSpace, FieldsPerRecord; LazyQuotes not
 * supported here), followed by the column selection of :1117-1131 (row[name] = line[index];
 * a missing field is "" when numFields < 0).  The Reader lives in the Go standard library (not
 * under /root/reference); restated from its documentation and source as remembered (go1.19+
 * encoding/csv/reader.go, readRecord/readLine):
 *   - input is consumed line by line; "\r\n" at a line end becomes "\n", a final "\r" at EOF is dropped
 *   - a line that is empty, or starts with the Comment rune, is skipped (only between records)
 *   - unquoted field: up to the next Comma; a '"' inside it is ErrBareQuote
 *   - quoted field: "" is a literal quote; the closing quote must be followed by Comma or the end
 *     of the line (else ErrQuote); it may span lines; EOF inside it is ErrQuote
 *   - FieldsPerRecord 0: every record must have as many fields as the first one; > 0: exactly
 *     that many; < 0: no check (ErrFieldCount otherwise)
 * Output: for `ncols` requested field indices, the values of the records after the first
 * `skip_records` ones, as SoA (two calls: sizes, then fill).                                   */
enum { ORC_CSV_OK = 0, ORC_CSV_ERR_BARE_QUOTE = 1, ORC_CSV_ERR_QUOTE = 2, ORC_CSV_ERR_FIELD_COUNT = 3 };

typedef struct {
    uint8_t comma, comment;          /* comment 0 = none */
    int32_t trim_leading_space;
    int32_t fields_per_record;       /* Go semantics */
    uint64_t skip_records;
} orc_csv_opts;

typedef struct {
    const uint8_t* d;
    uint64_t size, pos;
    /* current line (normalised): bytes [lb, le) of d, followed by a virtual '\n' if has_nl */
    uint64_t lb, le;
    int has_nl, eof;
} csv_lines;

static int csv_read_line(csv_lines* L) {   /* returns 0 at EOF (no more bytes) */
    if (L->pos >= L->size) { L->eof = 1; return 0; }
    uint64_t b = L->pos, e = b;
    while (e < L->size && L->d[e] != '\n') e++;
    L->has_nl = e < L->size;
    L->pos = L->has_nl ? e + 1 : e;
    if (L->has_nl) { if (e > b && L->d[e - 1] == '\r') e--; }        /* "\r\n" -> "\n" */
    else if (e > b && L->d[e - 1] == '\r') e--;                       /* final "\r" at EOF dropped */
    L->lb = b;
    L->le = e;
    return 1;
}

static uint64_t csv_trim_left(const uint8_t* d, uint64_t p, uint64_t e) {
    while (p < e) {
        if (!go_unicode_is_space_first_rune(d + p, e - p)) break;
        uint8_t b0 = d[p];
        p += b0 < 0x80 ? 1 : (b0 < 0xE0 ? 2 : 3);
    }
    return p;
}

/* sink: bytes of requested columns */
typedef struct {
    int32_t ncols;
    const int32_t* col_index;
    uint64_t* lens;        /* [ncols] running length of the current field (size pass) */
    uint8_t** out;         /* [ncols] write cursors (fill pass) or NULL */
} csv_sink;

static inline void sink_put(csv_sink* s, int field, uint8_t b) {
    for (int32_t c = 0; c < s->ncols; c++)
        if (s->col_index[c] == field) {
            if (s->out) *s->out[c]++ = b;
            s->lens[c]++;
        }
}

/* Reads one record; returns number of fields, or -1 at EOF; *err = kind. */
static int csv_read_record(csv_lines* L, const orc_csv_opts* o, csv_sink* s, int* err) {
    *err = 0;
    for (;;) {   /* skip empty and comment lines */
        if (!csv_read_line(L)) return -1;
        if (o->comment && L->le > L->lb && L->d[L->lb] == o->comment) continue;
        if (L->le == L->lb) continue;
        break;
    }
    const uint8_t* d = L->d;
    uint64_t p = L->lb;
    int field = 0;
    for (;;) {
        if (o->trim_leading_space) p = csv_trim_left(d, p, L->le);
        if (p >= L->le || d[p] != '"') {   /* unquoted */
            uint64_t i = p;
            while (i < L->le && d[i] != o->comma) {
                if (d[i] == '"' && !*err) *err = ORC_CSV_ERR_BARE_QUOTE;
                i++;
            }
            if (*err) return field + 1;
            for (uint64_t k = p; k < i; k++) sink_put(s, field, d[k]);
            field++;
            if (i < L->le) { p = i + 1; continue; }
            return field;
        }
        p++;   /* quoted */
        for (;;) {
            uint64_t i = p;
            while (i < L->le && d[i] != '"') i++;
            for (uint64_t k = p; k < i; k++) sink_put(s, field, d[k]);
            if (i < L->le) {   /* a quote */
                p = i + 1;
                if (p < L->le && d[p] == '"') { sink_put(s, field, '"'); p++; continue; }
                if (p < L->le && d[p] == o->comma) { p++; field++; break; }
                if (p == L->le) return field + 1;              /* closing quote at the end of the line */
                *err = ORC_CSV_ERR_QUOTE;
                return field + 1;
            }
            /* end of line inside the quoted field */
            if (L->has_nl) sink_put(s, field, '\n');
            if (!csv_read_line(L)) { *err = ORC_CSV_ERR_QUOTE; return field + 1; }   /* EOF inside quotes */
            p = L->lb;
        }
    }
}

/* Pass 1 (data == NULL in outs): returns the number of output records, fills col_bytes[ncols].
 * Pass 2: fills data/offsets (offsets: nrecords+1 uint64 per column).
 * On a parse error *err_kind != 0 and *err_record = 0-based index of the offending record (counting
 * every record read, header included); records before it are still produced. */
ORC_API uint64_t orc_csv_parse(const uint8_t* data, uint64_t size, const orc_csv_opts* o, const int32_t* col_index, int32_t ncols,
                               uint64_t* col_bytes, uint8_t** out_data, uint64_t** out_offsets, int32_t* err_kind,
                               uint64_t* err_record) {
    csv_lines L = {data, size, 0, 0, 0, 0, 0};
    uint64_t lens[64];
    uint8_t* cursors[64];
    uint64_t totals[64];
    csv_sink s = {ncols, col_index, lens, NULL};
    if (out_data) {
        for (int32_t c = 0; c < ncols; c++) cursors[c] = out_data[c];
        s.out = cursors;
    }
    for (int32_t c = 0; c < ncols; c++) totals[c] = 0;
    *err_kind = 0;
    *err_record = 0;
    uint64_t rec = 0, nout = 0;
    int expected = o->fields_per_record;
    for (;;) {
        for (int32_t c = 0; c < ncols; c++) lens[c] = 0;
        uint8_t* save[64];
        if (s.out) for (int32_t c = 0; c < ncols; c++) save[c] = cursors[c];
        int err = 0;
        int nf = csv_read_record(&L, o, &s, &err);
        if (nf < 0) break;
        if (!err) {
            if (expected == 0) expected = nf;
            else if (expected > 0 && nf != expected) err = ORC_CSV_ERR_FIELD_COUNT;
        }
        if (err) {
            if (s.out) for (int32_t c = 0; c < ncols; c++) cursors[c] = save[c];
            *err_kind = err;
            *err_record = rec;
            break;
        }
        if (rec >= o->skip_records) {
            for (int32_t c = 0; c < ncols; c++) {
                if (out_offsets) out_offsets[c][nout] = totals[c];
                totals[c] += lens[c];
            }
            nout++;
        } else if (s.out) {
            for (int32_t c = 0; c < ncols; c++) cursors[c] = save[c];   /* skipped record: drop its bytes */
        }
        rec++;
    }
    for (int32_t c = 0; c < ncols; c++) {
        if (out_offsets) out_offsets[c][nout] = totals[c];
        if (col_bytes) col_bytes[c] = totals[c];
    }
    return nout;
}

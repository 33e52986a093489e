// faithful.cpp — CPU baselines with the reference's COST MODEL (test / bench infrastructure, like the rest of oracle/).
//
// The C restatement next door (csvplus_oracle.c) answers "what must the result be"; it keeps strings in SoA
// columns and is therefore much leaner than the reference.  SURVEY.md §8(d) asks for two more timings beside it:
//
//   faithful   what csvplus.go actually does per row, single-threaded (there is no goroutine anywhere in it):
//              Row = map[string]string (csvplus.go:59) -> std::unordered_map<std::string, std::string> per row;
//              createIndex (:707-738) keeps pointers to the row maps and sorts them with a comparison sort whose
//              Less (:794-807) does one map lookup per side and column + strings.Compare; createUniqueIndex
//              (:740-756) scans adjacent rows with equalRows (:759-767); Join (:545-569) per stream row builds the
//              value slice (SelectValues :138-150), binary-searches with cmp (:893-920, map lookup per probe) and
//              allocates a merged map per match (mergeRows :571-583, stream value wins).  The chained join
//              orders.Join(customers, "cust_id").Join(products, "prod_id") is row-at-a-time like the reference's
//              nested closures.
//   lean-mt    the lean SoA algorithm (sort (key, row) pairs, binary-search probe) on ALL host cores (OpenMP,
//              libstdc++ parallel mode sort).
//
// Neither is the Go binary (no Go toolchain in this image): they are labelled "port" wherever they are reported.
// Nothing in the product path links or loads this file.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <parallel/algorithm>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {

typedef struct {
    const uint8_t* data;
    const void* offsets;
    uint64_t nrows;
    int32_t offset_bits;
    int32_t mem;
} fth_strcol;   // layout of orc_strcol / cph_strcol

}  // extern "C"

namespace {

using Row = std::unordered_map<std::string, std::string>;

inline uint64_t off_at(const fth_strcol& c, uint64_t i) {
    return c.offset_bits == 32 ? ((const uint32_t*)c.offsets)[i] : ((const uint64_t*)c.offsets)[i];
}
inline std::string value_of(const fth_strcol& c, uint64_t r) {
    const uint64_t b = off_at(c, r), e = off_at(c, r + 1);
    return std::string((const char*)c.data + b, (size_t)(e - b));
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Reader.Iterate builds one map per record (csvplus.go:1118-1131)
std::vector<Row> make_rows(const fth_strcol* cols, const char* const* names, int ncols) {
    const uint64_t n = cols[0].nrows;
    std::vector<Row> rows((size_t)n);
    for (uint64_t r = 0; r < n; r++) {
        Row& row = rows[(size_t)r];
        row.reserve((size_t)ncols);
        for (int c = 0; c < ncols; c++) row.emplace(names[c], value_of(cols[c], r));
    }
    return rows;
}

struct Index {
    std::vector<const Row*> rows;
    std::vector<std::string> columns;
};

// a missing key reads as "" (Go map semantics)
inline const std::string& get(const Row& r, const std::string& k) {
    static const std::string empty;
    auto it = r.find(k);
    return it == r.end() ? empty : it->second;
}

// indexImpl.Less :794-807
bool less_rows(const Index& ix, const Row* a, const Row* b) {
    for (const std::string& col : ix.columns) {
        const int c = get(*a, col).compare(get(*b, col));
        if (c < 0) return true;
        if (c > 0) return false;
    }
    return false;
}

// createIndex + createUniqueIndex: returns false on a duplicate
bool build_index(const std::vector<Row>& table, std::vector<std::string> columns, bool unique, Index* out) {
    out->columns = std::move(columns);
    out->rows.clear();
    for (const Row& r : table) out->rows.push_back(&r);                       // append, :729
    std::sort(out->rows.begin(), out->rows.end(), [&](const Row* a, const Row* b) { return less_rows(*out, a, b); });   // :736
    if (unique)
        for (size_t i = 1; i < out->rows.size(); i++) {                       // :749-753
            bool eq = true;
            for (const std::string& col : out->columns) eq = eq && get(*out->rows[i - 1], col) == get(*out->rows[i], col);
            if (eq) return false;
        }
    return true;
}

// cmp(i, values, eq) :907-920
inline bool cmp_at(const Index& ix, size_t i, const std::vector<std::string>& values, bool eq) {
    for (size_t j = 0; j < values.size(); j++) {
        const int c = get(*ix.rows[i], ix.columns[j]).compare(values[j]);
        if (c > 0) return true;
        if (c < 0) return false;
    }
    return eq;
}
// first(values) = sort.Search(n, cmp(i, values, true)) :893-897
inline size_t first_at(const Index& ix, const std::vector<std::string>& values) {
    size_t lo = 0, hi = ix.rows.size();
    while (lo < hi) {
        const size_t h = (lo + hi) >> 1;
        if (!cmp_at(ix, h, values, true)) lo = h + 1; else hi = h;
    }
    return lo;
}
// mergeRows :571-583: a fresh map, left (index row) then right (stream row): the stream's value wins
inline Row merge_rows(const Row& left, const Row& right) {
    Row r;
    r.reserve(left.size() + right.size());
    for (const auto& kv : left) r[kv.first] = kv.second;
    for (const auto& kv : right) r[kv.first] = kv.second;
    return r;
}
// Join closure body :553-567 for one stream row; fn receives every merged row
template <class F>
inline void join_row(const Index& ix, const std::vector<std::string>& columns, const Row& row, F&& fn) {
    std::vector<std::string> values;   // SelectValues :138-150: a fresh slice per row
    values.reserve(columns.size());
    for (const std::string& c : columns) values.push_back(get(row, c));
    for (size_t i = first_at(ix, values); i < ix.rows.size() && !cmp_at(ix, i, values, false); i++) fn(merge_rows(*ix.rows[i], row));
}

}  // namespace

extern "C" {

// orders.Join(customers by cust_key, orders.cust_col).Join(products by prod_key, orders.prod_col), everything as
// row maps.  times[0] = building the row maps of the two build tables (the Reader's work), [1] = the two
// UniqueIndexOn, [2] = row maps of the stream + the chained join.  Returns the joined-row count; `checksum`
// receives a value derived from every merged row so that nothing can be optimised away.
__attribute__((visibility("default"))) uint64_t fth_chain_join(const fth_strcol* cust, const char* const* cust_names, int cust_ncols,
                                                               const char* cust_key, const fth_strcol* prod,
                                                               const char* const* prod_names, int prod_ncols, const char* prod_key,
                                                               const fth_strcol* ords, const char* const* ord_names, int ord_ncols,
                                                               const char* ord_cust_col, const char* ord_prod_col, double* times,
                                                               uint64_t* checksum) {
    double t0 = now_s();
    std::vector<Row> crows = make_rows(cust, cust_names, cust_ncols), prows = make_rows(prod, prod_names, prod_ncols);
    times[0] = now_s() - t0;
    t0 = now_s();
    Index ci, pi;
    if (!build_index(crows, {cust_key}, true, &ci) || !build_index(prows, {prod_key}, true, &pi)) return UINT64_MAX;
    times[1] = now_s() - t0;
    t0 = now_s();
    uint64_t joined = 0, sum = 0;
    const std::vector<std::string> c1{ord_cust_col}, c2{ord_prod_col};
    const uint64_t m = ords[0].nrows;
    for (uint64_t r = 0; r < m; r++) {   // the stream: one row map at a time (csvplus.go:1118-1131), never materialised as a whole
        Row row;
        row.reserve((size_t)ord_ncols);
        for (int c = 0; c < ord_ncols; c++) row.emplace(ord_names[c], value_of(ords[c], r));
        join_row(ci, c1, row, [&](Row&& m1) {
            join_row(pi, c2, m1, [&](Row&& m2) {
                joined++;
                sum += m2.size();
                for (const auto& kv : m2) sum += kv.second.size();
            });
        });
    }
    times[2] = now_s() - t0;
    *checksum = sum;
    return joined;
}

// Lean SoA algorithm on `threads` cores (0 = all): unique index = parallel stable sort of (key, row), probe =
// parallel binary search.  build_row[r] = matching build row or 0xFFFFFFFF.  times[0] = sort, [1] = probe.
__attribute__((visibility("default"))) uint64_t fth_lean_mt_join(const fth_strcol* build, const fth_strcol* probe, int threads,
                                                                 uint32_t* build_row, double* times) {
    if (threads > 0) omp_set_num_threads(threads);
    const uint64_t n = build->nrows, m = probe->nrows;
    struct Key { const uint8_t* p; uint32_t len; uint32_t row; };
    auto key_of = [](const fth_strcol& c, uint64_t r) {
        const uint64_t b = off_at(c, r), e = off_at(c, r + 1);
        return Key{c.data + b, (uint32_t)(e - b), (uint32_t)r};
    };
    auto less = [](const Key& a, const Key& b) {
        const uint32_t k = a.len < b.len ? a.len : b.len;
        const int c = k ? memcmp(a.p, b.p, k) : 0;
        return c < 0 || (c == 0 && a.len < b.len);
    };
    double t0 = now_s();
    std::vector<Key> keys((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; r++) keys[(size_t)r] = key_of(*build, (uint64_t)r);
    __gnu_parallel::stable_sort(keys.begin(), keys.end(), less);
    times[0] = now_s() - t0;
    t0 = now_s();
    uint64_t joined = 0;
#pragma omp parallel for schedule(static) reduction(+ : joined)
    for (int64_t r = 0; r < (int64_t)m; r++) {
        const Key q = key_of(*probe, (uint64_t)r);
        size_t lo = 0, hi = (size_t)n;
        while (lo < hi) {
            const size_t h = (lo + hi) >> 1;
            if (less(keys[h], q)) lo = h + 1; else hi = h;
        }
        const bool hit = lo < (size_t)n && !less(q, keys[lo]);
        build_row[r] = hit ? keys[lo].row : 0xFFFFFFFFu;
        joined += hit;
    }
    times[1] = now_s() - t0;
    return joined;
}

__attribute__((visibility("default"))) int fth_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"

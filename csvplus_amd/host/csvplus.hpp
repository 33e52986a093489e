// csvplus.hpp — C++ host side of the hot path, mirroring the reference's Go method set
// (maxim2266/csvplus, csvplus.go) for Index build + Join on top of the C ABI
// (include/csvplus_hip.h).  The Go toolchain is absent from this image, so this header is
// what the cgo shim of INTEGRATION.md would be in Go: same names, argument meaning and
// error behaviour.
//
//   Go (csvplus.go)                              here
//   type Row map[string]string            :59    csvplus::Row
//   Row.HasColumn/SelectValues/...      :62-161  free functions of the same names
//   type DataSource func(RowFunc) error  :215    csvplus::DataSource
//   TakeRows / Take                    :218,:252 csvplus::TakeRows / Take
//   DataSource.IndexOn / UniqueIndexOn :529,:535 DataSource::IndexOn / UniqueIndexOn
//   DataSource.Join / Except           :545,:588 DataSource::Join / Except
//   Index.Iterate / Find / SubIndex    :618-641  Index::Iterate / Find / SubIndex
//   Index.ResolveDuplicates            :643-653  Index::ResolveDuplicates (groups found on the GPU)
//   Index.WriteTo / LoadIndex          :655-705  Index::WriteTo / LoadIndex (own binary format, not gob)
//   DataSourceError                   :1229-1238 csvplus::DataSourceError
//
// Go `error` values become csvplus::Error (nil == ok()); Go panics (programmer errors:
// empty / duplicate column lists, too many join columns, :710, :715, :549, :634) become
// csvplus::Panic exceptions.  io.EOF keeps its role: returned from a RowFunc it stops the
// iteration cleanly (:213-214, :238-239).
//
// What runs where: rows stay host-side maps exactly as in the reference; the key columns of
// a batch are staged into pinned SoA buffers and the sort / unique check / probe run on the
// GPU.  There is no CPU implementation of those steps here.
//
// Chained Joins (round 4): `src.Join(a, ka).Join(b, kb)...` (README.md:56; csvplus.go:545-569 nested: the second Join's
// source IS the first Join's closure) is recognised — a DataSource returned by Join remembers its upstream and its steps
// — and a batch of stream rows goes through ONE cph_join_chain_ex(CPH_CHAIN_POSITIONS) call, the entry point bench.py
// times, as long as every row of the batch carries the later steps' key columns itself: mergeRows
// lets the stream's value win, so that is the value the later Join sees.  Round 5: a later key that NO stream row of the
// batch carries but every row of an earlier index does — people.Join(orders, "id").DropColumns(...).Join(products),
// csvplus_test.go:280-285: prod_id is a column of the orders index rows — is fused too: the step names its source
// (cph_chain_step.source) and the device gathers the key from the row that earlier step matched.  DropColumns between the
// Joins of a chain is part of the chain (the columns are dropped when the merged row is built).  A batch whose rows
// disagree about where a key comes from runs the steps one after the other over the merged rows (same emission order,
// same errors).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "csvplus_hip.h"

namespace csvplus {

// ---- errors ------------------------------------------------------------------------------------
class Error {
public:
    Error() = default;                                        // nil
    explicit Error(std::string msg) : set_(true), msg_(std::move(msg)) {}
    static Error Eof() { Error e("EOF"); e.eof_ = true; return e; }   // io.EOF
    bool ok() const { return !set_; }
    bool is_eof() const { return set_ && eof_; }
    explicit operator bool() const { return set_; }           // `if (err)` == `if err != nil`
    const std::string& message() const { return msg_; }       // err.Error()
    // DataSourceError (:1229-1238)
    bool is_data_source_error() const { return has_line_; }
    uint64_t line() const { return line_; }
    static Error DataSourceError(uint64_t line, const Error& inner) {
        Error e("row " + std::to_string(line) + ": " + inner.message());
        e.has_line_ = true;
        e.line_ = line;
        return e;
    }

private:
    bool set_ = false, eof_ = false, has_line_ = false;
    uint64_t line_ = 0;
    std::string msg_;
};
inline const Error io_EOF = Error::Eof();

struct Panic : std::logic_error {   // Go panic(...)
    using std::logic_error::logic_error;
};

// ---- Row (:59) -----------------------------------------------------------------------------------
using Row = std::map<std::string, std::string>;

inline bool HasColumn(const Row& row, const std::string& col) { return row.count(col) != 0; }   // :62-65

inline std::string quote(const std::string& s) { return "\"" + s + "\""; }   // %q for plain names

// SelectValues (:138-150)
inline Error SelectValues(const Row& row, const std::vector<std::string>& cols, std::vector<const std::string*>* out) {
    out->resize(cols.size());
    for (size_t i = 0; i < cols.size(); i++) {
        auto it = row.find(cols[i]);
        if (it == row.end()) return Error("missing column " + quote(cols[i]));
        (*out)[i] = &it->second;
    }
    return Error();
}

// SelectExisting (:108-118)
inline Row SelectExisting(const Row& row, const std::vector<std::string>& cols) {
    Row r;
    for (const auto& name : cols) {
        auto it = row.find(name);
        if (it != row.end()) r[name] = it->second;
    }
    return r;
}

// Row.String (:90-104): `{ "a" : "1", "b" : "2" }`, columns sorted
inline std::string String(const Row& row) {
    if (row.empty()) return "{}";
    std::string s = "{ ";
    bool first = true;
    for (const auto& kv : row) {   // std::map iterates in sorted key order == Header()
        if (!first) s += ", ";
        first = false;
        s += "\"" + kv.first + "\" : \"" + kv.second + "\"";
    }
    return s + " }";
}

// mergeRows (:571-583): index row first, then the stream row: the stream wins on a name collision
inline Row mergeRows(const Row& left, const Row& right) {
    Row r = left;
    for (const auto& kv : right) r[kv.first] = kv.second;
    return r;
}

using RowFunc = std::function<Error(Row)>;

// ---- GPU context shared by the process (the Go API has no context argument) -------------------------
class Gpu {
public:
    static Gpu& Default() {
        static Gpu g;
        return g;
    }
    cph_ctx* ctx() {
        if (!ctx_) {
            int dev = 0;
            if (const char* e = std::getenv("CSVPLUS_HIP_DEVICE")) dev = std::atoi(e);
            int32_t rc = cph_ctx_create(dev, &ctx_);
            if (rc != CPH_OK) throw std::runtime_error("csvplus: no usable GPU (cph_ctx_create failed, status " +
                                                       std::to_string(rc) + "); there is no CPU fallback");
        }
        return ctx_;
    }
    // The process-wide ctx is deliberately never destroyed: indexes held in static storage may
    // outlive it during exit, and they return their device blocks to this ctx's pool.
    // Rows read ahead per GPU probe call in Join/Except.  1 reproduces the reference's
    // row-at-a-time order of side effects exactly.
    size_t join_batch_rows = 8192;

private:
    cph_ctx* ctx_ = nullptr;
};

namespace detail {

// Key columns of a batch of rows, staged as Arrow-style SoA in pinned host memory.
class StagedColumns {
public:
    StagedColumns(cph_ctx* ctx, size_t ncols) : ctx_(ctx), data_(ncols), offs_(ncols), cols_(ncols) {}
    StagedColumns(const StagedColumns&) = delete;
    ~StagedColumns() {
        for (void* p : pinned_) cph_pinned_free(ctx_, p);
    }
    // values[c][i] = value of key column c in row i
    void stage(const std::vector<std::vector<const std::string*>>& values, size_t nrows) {
        for (size_t c = 0; c < cols_.size(); c++) {
            uint64_t total = 0;
            for (size_t i = 0; i < nrows; i++) total += values[c][i]->size();
            uint8_t* d = static_cast<uint8_t*>(pinned(total + 8));
            uint64_t* o = static_cast<uint64_t*>(pinned((nrows + 1) * sizeof(uint64_t)));
            uint64_t pos = 0;
            for (size_t i = 0; i < nrows; i++) {
                o[i] = pos;
                const std::string& v = *values[c][i];
                if (!v.empty()) std::memcpy(d + pos, v.data(), v.size());
                pos += v.size();
            }
            o[nrows] = pos;
            cols_[c].data = d;
            cols_[c].offsets = o;
            cols_[c].nrows = nrows;
            cols_[c].offset_bits = 64;
            cols_[c].mem = CPH_MEM_HOST;
        }
    }
    const cph_strcol* cols() const { return cols_.data(); }

private:
    void* pinned(size_t bytes) {
        void* p = nullptr;
        if (cph_pinned_alloc(ctx_, bytes, &p) != CPH_OK)
            throw std::runtime_error(std::string("csvplus: ") + cph_last_error(ctx_));
        pinned_.push_back(p);
        return p;
    }
    cph_ctx* ctx_;
    std::vector<void*> data_, offs_;
    std::vector<cph_strcol> cols_;
    std::vector<void*> pinned_;
};

struct DeviceIndex {   // owns a cph_index
    cph_index* h = nullptr;
    ~DeviceIndex() {
        if (h) cph_index_destroy(h);
    }
};

// allColumnsUnique (:770-782)
inline bool allColumnsUnique(const std::vector<std::string>& columns) {
    std::set<std::string> s(columns.begin(), columns.end());
    return s.size() == columns.size();
}

}  // namespace detail

class DataSource;

// ---- Index (:612-641) ------------------------------------------------------------------------------
class Index {
public:
    // Iterate (:618-620): rows sorted on the index columns, each handed out as a copy (:230)
    Error Iterate(const RowFunc& fn) const;
    // Find (:625-627)
    DataSource Find(const std::vector<std::string>& values) const;
    // SubIndex (:632-641)
    std::shared_ptr<Index> SubIndex(const std::vector<std::string>& values) const;

    // ResolveDuplicates (:643-653): `resolve` is called once per pack of rows with equal key, in key order;
    // it returns one of the rows (kept as the only row with that key), an empty row (the pack is dropped) or an
    // error (returned to the caller).  Compaction follows dedup (:810-867) to the letter.
    using ResolveFunc = std::function<std::pair<Row, Error>(const std::vector<Row>&)>;
    Error ResolveDuplicates(const ResolveFunc& resolve);
    // WriteTo (:655-680) / LoadIndex (:682-705).  The reference gob-encodes columns + rows; this file is a
    // length-prefixed little-endian dump of the same two values and is NOT readable by the Go library.
    Error WriteTo(const std::string& fileName) const;

    const std::vector<Row>& rows() const { return impl_rows; }
    const std::vector<std::string>& columns() const { return impl_columns; }

    // ---- indexImpl (:785-788): the sorted rows ARE the index ----
    std::vector<Row> impl_rows;
    std::vector<std::string> impl_columns;

    // find (:870-891): [lower, upper) over impl_rows
    std::pair<size_t, size_t> find(const std::vector<std::string>& values) const {
        if (values.empty()) return {0, impl_rows.size()};                                  // :872-874
        if (values.size() > impl_columns.size()) throw Panic("too many columns in indexImpl.find()");   // :876-878
        std::vector<cph_strval> v(values.size());
        for (size_t i = 0; i < values.size(); i++) {
            v[i].data = reinterpret_cast<const uint8_t*>(values[i].data());
            v[i].len = values[i].size();
        }
        uint64_t lo = 0, hi = 0;
        cph_ctx* ctx = Gpu::Default().ctx();
        if (cph_index_find(ctx, device().h, v.data(), (int32_t)v.size(), &lo, &hi) != CPH_OK)
            throw std::runtime_error(std::string("csvplus: ") + cph_last_error(ctx));
        return {(size_t)lo, (size_t)hi};
    }

    // The GPU-resident twin of impl_rows' key columns; built on demand for sub-indices.
    detail::DeviceIndex& device() const {
        if (!dev_) {
            dev_ = std::make_shared<detail::DeviceIndex>();
            cph_ctx* ctx = Gpu::Default().ctx();
            std::vector<std::vector<const std::string*>> vals(impl_columns.size());
            for (size_t c = 0; c < impl_columns.size(); c++) {
                vals[c].resize(impl_rows.size());
                for (size_t i = 0; i < impl_rows.size(); i++) vals[c][i] = &impl_rows[i].at(impl_columns[c]);
            }
            detail::StagedColumns st(ctx, impl_columns.size());
            st.stage(vals, impl_rows.size());
            uint64_t dup = 0;
            int32_t rc = cph_index_build(ctx, st.cols(), (int32_t)impl_columns.size(), 0, &dev_->h, &dup);
            if (rc != CPH_OK) throw std::runtime_error(std::string("csvplus: ") + cph_last_error(ctx));
        }
        return *dev_;
    }
    mutable std::shared_ptr<detail::DeviceIndex> dev_;

    // A column of impl_rows as pinned SoA in SORTED order (row i = impl_rows[i]), staged once: what a later Join of a chain reads
    // its key from when the key is a column of THIS index's rows (cph_chain_step.source < 0).  nullptr when some row lacks it.
    // How many rows of impl_rows carry column `name`: kAll / kNone / kSome — ONE pass over the rows per column, cached until the rows
    // change (a chain asks per batch of stream rows: at 1e7 index rows the uncached walk dwarfed the 60-80 us device call)
    enum Presence { kNone = 0, kSome = 1, kAll = 2 };
    Presence column_presence(const std::string& name) const {
        auto it = col_presence_.find(name);
        if (it != col_presence_.end()) return it->second;
        size_t have = 0;
        for (const Row& r : impl_rows) have += r.count(name) ? 1 : 0;
        Presence p = have == 0 ? kNone : have == impl_rows.size() ? kAll : kSome;
        col_presence_.emplace(name, p);
        return p;
    }
    const cph_strcol* side_column(cph_ctx* ctx, const std::string& name) const {
        auto it = side_cols_.find(name);
        if (it == side_cols_.end()) {
            std::shared_ptr<detail::StagedColumns> st;
            const bool all = column_presence(name) == kAll;
            if (all) {
                std::vector<std::vector<const std::string*>> vals(1);
                vals[0].resize(impl_rows.size());
                for (size_t i = 0; i < impl_rows.size(); i++) vals[0][i] = &impl_rows[i].at(name);
                st = std::make_shared<detail::StagedColumns>(ctx, 1);
                st->stage(vals, impl_rows.size());
            }
            it = side_cols_.emplace(name, std::move(st)).first;
        }
        return it->second ? it->second->cols() : nullptr;
    }
    void invalidate_device() {   // impl_rows changed (ResolveDuplicates): the device twin and the staged columns are rebuilt on next use
        dev_.reset();
        side_cols_.clear();
        col_presence_.clear();
    }
    mutable std::map<std::string, std::shared_ptr<detail::StagedColumns>> side_cols_;
    mutable std::map<std::string, Presence> col_presence_;
};

// ---- DataSource (:215) ---------------------------------------------------------------------------------
class DataSource {
public:
    using Fn = std::function<Error(const RowFunc&)>;
    DataSource() = default;
    DataSource(Fn f) : fn_(std::move(f)) {}   // NOLINT: implicit like a Go func value
    Error operator()(const RowFunc& fn) const { return fn_(fn); }

    // IndexOn (:529-531) / UniqueIndexOn (:535-537): (index, error); index is null on error
    std::pair<std::shared_ptr<Index>, Error> IndexOn(const std::vector<std::string>& columns) const {
        return createIndex(columns, false);
    }
    std::pair<std::shared_ptr<Index>, Error> UniqueIndexOn(const std::vector<std::string>& columns) const {
        return createIndex(columns, true);
    }

    // Join (:545-569).  `columns` empty = natural join on the index columns.
    DataSource Join(std::shared_ptr<Index> index, std::vector<std::string> columns = {}) const {
        if (columns.empty()) columns = index->impl_columns;                                  // :546-547
        else if (columns.size() > index->impl_columns.size()) throw Panic("too many source columns in Join()");  // :548-550
        // a Join over a Join: one more step of the same chain (at most CPH_MAX_CHAIN fused; a longer chain starts a new one)
        auto spec = std::make_shared<ChainSpec>();
        if (chain_ && chain_->steps.size() < (size_t)CPH_MAX_CHAIN) *spec = *chain_;
        else spec->upstream = fn_;
        spec->steps.push_back(ChainStepSpec{std::move(index), std::move(columns)});
        DataSource out = spec->steps.size() == 1 ? probeSource(spec->upstream, spec->steps[0].index, spec->steps[0].columns, /*anti=*/false)
                                                 : chainSource(spec);   // (a one-step chain gets its drops in DropColumns)
        out.chain_ = spec;
        return out;
    }

    // Except (:588-608)
    DataSource Except(std::shared_ptr<Index> index, std::vector<std::string> columns = {}) const {
        if (columns.empty()) columns = index->impl_columns;
        else if (columns.size() > index->impl_columns.size()) throw Panic("too many source columns in Except()");
        return probeSource(fn_, std::move(index), std::move(columns), /*anti=*/true);
    }

    // DropColumns (:493-509).  Between the Joins of a chain it stays part of the chain: the columns leave the merged row when
    // it is built, and a later Join that needs one of them fails as in the reference (missing column).
    DataSource DropColumns(std::vector<std::string> columns) const {
        if (columns.empty()) throw Panic("no columns specified in DropColumns()");
        if (chain_) {
            auto spec = std::make_shared<ChainSpec>(*chain_);
            auto& d = spec->steps.back().drop_after;
            d.insert(d.end(), columns.begin(), columns.end());
            DataSource out = chainSource(spec);
            out.chain_ = spec;
            return out;
        }
        Fn src = fn_;
        return DataSource([src, columns](const RowFunc& fn) {
            return src([&](Row row) {
                for (const auto& c : columns) row.erase(c);
                return fn(std::move(row));
            });
        });
    }
    // SelectColumns (:511-525)
    DataSource SelectColumns(std::vector<std::string> columns) const {
        if (columns.empty()) throw Panic("no columns specified in SelectColumns()");
        Fn src = fn_;
        return DataSource([src, columns](const RowFunc& fn) {
            return src([&](Row row) {
                Row r;
                for (const auto& c : columns) {
                    auto it = row.find(c);
                    if (it == row.end()) return Error("missing column " + quote(c));   // Row.Select :122-135
                    r[c] = it->second;
                }
                return fn(std::move(r));
            });
        });
    }

    // ToRows (:481-490)
    std::pair<std::vector<Row>, Error> ToRows() const {
        std::vector<Row> rows;
        Error err = fn_([&](Row row) { rows.push_back(std::move(row)); return Error(); });
        return {std::move(rows), err};
    }

private:
    Fn fn_;
    struct ChainStepSpec {
        std::shared_ptr<Index> index;
        std::vector<std::string> columns;
        std::vector<std::string> drop_after;   // DropColumns applied to this step's output rows
    };
    struct ChainSpec {   // this source == upstream.Join(steps[0])...Join(steps.back())
        Fn upstream;
        std::vector<ChainStepSpec> steps;
    };
    std::shared_ptr<const ChainSpec> chain_;

    // createIndex (:707-738) + createUniqueIndex (:740-756)
    std::pair<std::shared_ptr<Index>, Error> createIndex(const std::vector<std::string>& columns, bool unique) const {
        if (columns.empty()) throw Panic("empty column list in CreateIndex()");                    // :709-710
        if (columns.size() > 1 && !detail::allColumnsUnique(columns))
            throw Panic("duplicate column name(s) in CreateIndex()");                                // :714-716
        auto index = std::make_shared<Index>();
        index->impl_columns = columns;
        std::vector<Row> rows;
        // copy Rows with validation (:722-733)
        Error err = fn_([&](Row row) {
            for (const auto& col : columns)
                if (!HasColumn(row, col)) return Error("missing column " + quote(col) + " while creating an index");
            rows.push_back(std::move(row));
            return Error();
        });
        if (err) return {nullptr, err};
        if (rows.size() > 0xFFFFFFFFull) return {nullptr, Error("too many rows for a GPU index (2^32-1 max)")};

        // sort (:736) on the GPU: stage key columns, build, fetch the permutation
        cph_ctx* ctx = Gpu::Default().ctx();
        std::vector<std::vector<const std::string*>> vals(columns.size());
        for (size_t c = 0; c < columns.size(); c++) {
            vals[c].resize(rows.size());
            for (size_t i = 0; i < rows.size(); i++) vals[c][i] = &rows[i].at(columns[c]);
        }
        auto dev = std::make_shared<detail::DeviceIndex>();
        uint64_t first_dup = UINT64_MAX;
        int32_t rc;
        {
            detail::StagedColumns st(ctx, columns.size());
            st.stage(vals, rows.size());
            rc = cph_index_build(ctx, st.cols(), (int32_t)columns.size(), unique ? 1 : 0, &dev->h, &first_dup);
        }
        if (rc != CPH_OK && rc != CPH_ERR_DUPLICATE) return {nullptr, Error(std::string("csvplus_hip: ") + cph_last_error(ctx))};
        const uint32_t* perm = nullptr;
        uint64_t n = 0;
        if (cph_index_perm(dev->h, CPH_MEM_HOST, &perm, &n) != CPH_OK)
            return {nullptr, Error(std::string("csvplus_hip: ") + cph_last_error(ctx))};
        if (rc == CPH_ERR_DUPLICATE) {
            // :751  "duplicate value while creating unique index: " + rows[i].SelectExisting(columns...).String()
            return {nullptr, Error("duplicate value while creating unique index: " +
                                   String(SelectExisting(rows[perm[first_dup]], columns)))};
        }
        index->impl_rows.resize(rows.size());
        for (size_t i = 0; i < rows.size(); i++) index->impl_rows[i] = std::move(rows[perm[i]]);
        index->dev_ = dev;
        return {index, Error()};
    }

    // One step over a batch of rows that all carry `columns`: the per-row `first()` + forward scan (:557-563) / `has()`
    // (:599-602) as one GPU probe; `emit` gets every output row in the reference's order.  The batch is consumed.
    static Error probeBatch(cph_ctx* ctx, const std::shared_ptr<Index>& index, const std::vector<std::string>& columns, bool anti,
                            std::vector<Row>* batch, const RowFunc& emit) {
        if (batch->empty()) return Error();
        std::vector<std::vector<const std::string*>> vals(columns.size());
        for (auto& v : vals) v.resize(batch->size());
        for (size_t i = 0; i < batch->size(); i++)
            for (size_t c = 0; c < columns.size(); c++) vals[c][i] = &(*batch)[i].at(columns[c]);
        cph_matches* m = nullptr;
        {
            detail::StagedColumns st(ctx, columns.size());
            st.stage(vals, batch->size());
            int32_t rc = cph_join_probe(ctx, index->device().h, st.cols(), (int32_t)columns.size(), nullptr, 32,
                                        0, 0, 0, /*want_pairs=*/0, CPH_MEM_HOST, &m);
            if (rc != CPH_OK) return Error(std::string("csvplus_hip: ") + cph_last_error(ctx));
        }
        Error err;
        // index.impl.rows is read live at iteration time (:557), as in the reference
        const std::vector<Row>& irows = index->impl_rows;
        for (size_t i = 0; i < batch->size() && !err; i++) {
            if (anti) {
                if (m->cnt[i] == 0) err = emit(std::move((*batch)[i]));                        // :600-602
            } else {
                for (uint32_t j = 0; j < m->cnt[i] && !err; j++)                              // :559-563
                    err = emit(mergeRows(irows[(size_t)m->lo[i] + j], (*batch)[i]));
            }
        }
        cph_matches_release(m);
        batch->clear();
        return err;
    }

    // Reads `src` in batches of Gpu::join_batch_rows rows that carry `columns` (SelectValues, :556 / :599: a row without one
    // of them ends the iteration with that error — after the rows in front of it went through `flush`) and hands each batch
    // to `flush`.
    static Error batched(const Fn& src, const std::vector<std::string>& columns, const std::function<Error(std::vector<Row>*)>& flush) {
        const size_t batch_rows = std::max<size_t>(1, Gpu::Default().join_batch_rows);
        std::vector<Row> batch;
        batch.reserve(batch_rows);
        Error err = src([&](Row row) -> Error {
            std::vector<const std::string*> tmp;
            Error e = SelectValues(row, columns, &tmp);                                       // :556 / :599
            if (e) {
                // rows before this one must be delivered first, then the error surfaces
                Error fe = flush(&batch);
                return fe ? fe : e;
            }
            batch.push_back(std::move(row));
            if (batch.size() >= batch_rows) return flush(&batch);
            return Error();
        });
        if (err) return err;
        err = flush(&batch);
        // an io.EOF from `fn` in the tail batch ends the iteration cleanly, as the source
        // would have mapped it (:238-239)
        return err.is_eof() ? Error() : err;
    }

    // Shared body of a single Join / Except over `src`.
    static DataSource probeSource(Fn src, std::shared_ptr<Index> index, std::vector<std::string> columns, bool anti) {
        return DataSource([src, index, columns, anti](const RowFunc& fn) -> Error {
            cph_ctx* ctx = Gpu::Default().ctx();
            return batched(src, columns, [&](std::vector<Row>* batch) { return probeBatch(ctx, index, columns, anti, batch, fn); });
        });
    }

    // upstream.Join(steps[0])...Join(steps[k]) (with the DropColumns between them).  Per batch of stream rows: ONE fused device
    // call — cph_join_chain_ex reporting sorted positions, the subscripts into each index's impl_rows — when every later key
    // has ONE origin for the whole batch: every stream row carries it (the value a later Join sees is then the stream's own:
    // mergeRows :571-583 lets the right operand win), or no stream row does and every row of an earlier index does
    // (cph_chain_step.source: the device reads the key from the row that step matched; the earliest such index wins, as in the
    // nested merges).  Else the steps one after the other over the merged rows.  Either way the rows come out in the reference's
    // nested order: by stream row, then by position in steps[0]'s index, then in steps[1]'s, ...; a step-k row is merged as
    // mergeRows(index_k row, mergeRows(index_k-1 row, ... stream row)): on a shared column name the precedence is
    // stream > steps[0] > steps[1] > ... (csvplus.go:559-560 nested).
    static DataSource chainSource(std::shared_ptr<const ChainSpec> spec) {
        return DataSource([spec](const RowFunc& fn) -> Error {
            cph_ctx* ctx = Gpu::Default().ctx();
            const size_t S = spec->steps.size();
            return batched(spec->upstream, spec->steps[0].columns, [&](std::vector<Row>* batch) -> Error {
                if (batch->empty()) return Error();
                // origin[k]: 0 = step k's key columns come from the stream rows, t + 1 = from the rows of steps[t].index
                std::vector<int> origin(S, 0);
                bool fusable = true;
                for (size_t k = 1; k < S && fusable; k++) {
                    int org = -2;   // not decided
                    for (const auto& col : spec->steps[k].columns) {
                        const int o = columnOrigin(ctx, *spec, k, col, *batch);
                        if (o < 0 || (org != -2 && o != org)) { fusable = false; break; }
                        org = o;
                    }
                    origin[k] = org;
                }
                // the steps one after the other: step k's output rows are step k+1's stream (same order of outputs and of
                // errors as the nested closures: a later step sees the rows in emission order)
                if (!fusable) return runSteps(ctx, *spec, 0, batch, fn);
                // ---- fused: one device call ----
                std::vector<std::unique_ptr<detail::StagedColumns>> staged;
                std::vector<std::vector<cph_strcol>> side(S);
                std::vector<cph_chain_step> steps(S);
                for (size_t k = 0; k < S; k++) {
                    const auto& cols = spec->steps[k].columns;
                    steps[k] = cph_chain_step{};
                    steps[k].index = spec->steps[k].index->device().h;
                    steps[k].ncols = (int32_t)cols.size();
                    if (origin[k] == 0) {
                        std::vector<std::vector<const std::string*>> vals(cols.size());
                        for (auto& v : vals) v.resize(batch->size());
                        for (size_t i = 0; i < batch->size(); i++)
                            for (size_t c = 0; c < cols.size(); c++) vals[c][i] = &(*batch)[i].at(cols[c]);
                        staged.emplace_back(new detail::StagedColumns(ctx, cols.size()));
                        staged.back()->stage(vals, batch->size());
                        steps[k].cols = staged.back()->cols();
                    } else {   // the index's own rows, in sorted order (impl_rows), staged once per index
                        const Index& from = *spec->steps[(size_t)origin[k] - 1].index;
                        (void)from.device();   // positions must refer to impl_rows as they are now
                        for (const auto& c : cols) side[k].push_back(*from.side_column(ctx, c));
                        steps[k].cols = side[k].data();
                        steps[k].source = -origin[k];
                    }
                }
                cph_chain* ch = nullptr;
                if (cph_join_chain_ex(ctx, steps.data(), (int32_t)S, 0, CPH_MEM_HOST, CPH_CHAIN_POSITIONS, &ch) != CPH_OK)
                    return Error(std::string("csvplus_hip: ") + cph_last_error(ctx));
                fused_calls()++;
                Error err;
                for (uint64_t m = 0; m < ch->nrows && !err; m++) {
                    const uint64_t r = ch->stream_row ? ch->stream_row[m] : m;   // NULL: every row joined exactly once, in order
                    Row row = (*batch)[(size_t)r];
                    for (size_t k = 0; k < S; k++) {
                        row = mergeRows(spec->steps[k].index->impl_rows[ch->build_row[k][m]], row);
                        for (const auto& c : spec->steps[k].drop_after) row.erase(c);
                    }
                    err = fn(std::move(row));
                }
                cph_chain_release(ch);
                batch->clear();
                return err;
            });
        });
    }

    // Where the value of `col` in the row that step k's Join sees comes from, for a whole batch: 0 = every stream row carries
    // it (and no DropColumns in between removed it), t + 1 = no stream row does and every row of steps[t].index does (the
    // earliest such t < k: mergeRows lets the earlier, right-hand row win), -1 = the rows disagree or the column is gone.
    static int columnOrigin(cph_ctx* ctx, const ChainSpec& spec, size_t k, const std::string& col, const std::vector<Row>& batch) {
        auto dropped_between = [&](size_t from_step) {   // a DropColumns behind steps[from_step .. k-1] names the column
            for (size_t j = from_step; j < k; j++)
                for (const auto& d : spec.steps[j].drop_after)
                    if (d == col) return true;
            return false;
        };
        size_t have = 0;
        for (const Row& r : batch) have += HasColumn(r, col) ? 1 : 0;
        if (have == batch.size()) return dropped_between(0) ? -1 : 0;
        if (have != 0) return -1;
        for (size_t t = 0; t < k; t++) {
            const Index& ix = *spec.steps[t].index;
            const Index::Presence p = ix.column_presence(col);   // cached per (index, column): no walk over the rows per batch
            if (p == Index::kSome) return -1;                    // some rows of this index carry it, some do not
            if (p == Index::kAll) return ix.side_column(ctx, col) && !dropped_between(t) ? (int)t + 1 : -1;
        }
        return -1;
    }

public:
    // number of fused device calls made so far (tests: one call per batch)
    static uint64_t& fused_calls() {
        static uint64_t n = 0;
        return n;
    }

private:
    // steps[from..] one after the other over `rows` (rows of step `from`'s stream that all carry its key columns)
    static Error runSteps(cph_ctx* ctx, const ChainSpec& spec, size_t from, std::vector<Row>* rows, const RowFunc& fn) {
        std::vector<Row> cur = std::move(*rows);
        for (size_t k = from; k < spec.steps.size(); k++) {
            const bool last = k + 1 == spec.steps.size();
            const auto& st = spec.steps[k];
            if (k > from) {
                size_t good = 0;
                Error miss;
                for (; good < cur.size(); good++) {
                    std::vector<const std::string*> tmp;
                    miss = SelectValues(cur[good], st.columns, &tmp);
                    if (miss) break;
                }
                if (miss) {
                    cur.resize(good);
                    Error e = runSteps(ctx, spec, k, &cur, fn);
                    return e ? e : miss;
                }
            }
            std::vector<Row> next;
            const RowFunc sink = last ? fn : RowFunc([&](Row row) { next.push_back(std::move(row)); return Error(); });
            Error e = probeBatch(ctx, st.index, st.columns, false, &cur,
                                 st.drop_after.empty() ? sink : RowFunc([&](Row row) {
                                     for (const auto& c : st.drop_after) row.erase(c);
                                     return sink(std::move(row));
                                 }));
            if (e) return e;
            cur = std::move(next);
        }
        return Error();
    }
};

// iterate (:225-249): clones each row, io.EOF -> nil, other errors -> DataSourceError{Line: i}
inline Error iterate(const std::vector<Row>& rows, const RowFunc& fn) {
    Error err;
    size_t i = 0;
    for (; i < rows.size(); i++) {
        err = fn(rows[i]);   // pass-by-value parameter: the callee gets a copy (Clone, :230)
        if (err) break;
    }
    if (!err) return err;
    if (err.is_eof()) return Error();
    return Error::DataSourceError((uint64_t)i, err);
}

// TakeRows (:218-222).  The rows are captured by value (a Go slice header copy shares the
// backing array; here the DataSource owns a snapshot).
inline DataSource TakeRows(std::vector<Row> rows) {
    auto shared = std::make_shared<std::vector<Row>>(std::move(rows));
    return DataSource([shared](const RowFunc& fn) { return iterate(*shared, fn); });
}

// Take (:252-256): anything with Iterate(fn)
template <class T>
inline DataSource Take(std::shared_ptr<T> src) {
    return DataSource([src](const RowFunc& fn) { return src->Iterate(fn); });
}

inline Error Index::Iterate(const RowFunc& fn) const { return iterate(impl_rows, fn); }

inline DataSource Index::Find(const std::vector<std::string>& values) const {
    auto r = find(values);
    return TakeRows(std::vector<Row>(impl_rows.begin() + (long)r.first, impl_rows.begin() + (long)r.second));
}

inline std::shared_ptr<Index> Index::SubIndex(const std::vector<std::string>& values) const {
    if (values.size() >= impl_columns.size()) throw Panic("too many values in SubIndex()");   // :633-635
    auto r = find(values);
    auto sub = std::make_shared<Index>();
    sub->impl_rows.assign(impl_rows.begin() + (long)r.first, impl_rows.begin() + (long)r.second);
    sub->impl_columns.assign(impl_columns.begin() + (long)values.size(), impl_columns.end());
    return sub;
}

// dedup (:810-867).  The adjacent-equal scans (:815-819, :851-855) and the per-group binary search (:828-830)
// are one GPU pass over the sorted key codes (cph_index_dup_groups); the loop below replays the reference's
// bookkeeping over that list of groups, so the surviving rows — including the reference's habit of not
// copying the final row when the last pack ends before it (:851-859 moves rows[lower-1] only while
// lower < len) — are exactly the reference's.
inline Error Index::ResolveDuplicates(const ResolveFunc& resolve) {
    cph_ctx* ctx = Gpu::Default().ctx();
    cph_groups* g = nullptr;
    if (cph_index_dup_groups(ctx, device().h, &g) != CPH_OK)
        throw std::runtime_error(std::string("csvplus: ") + cph_last_error(ctx));
    struct Release { cph_groups* g; ~Release() { cph_groups_release(g); } } rel{g};
    if (g->ngroups == 0) return Error();                                            // :821-823
    const size_t n = impl_rows.size();
    size_t dest = (size_t)g->lower[0];                                              // dest = lower-1 (:825)
    for (uint64_t k = 0; k < g->ngroups; k++) {
        const size_t lo = (size_t)g->lower[k], hi = (size_t)g->upper[k];
        std::vector<Row> pack(impl_rows.begin() + (long)lo, impl_rows.begin() + (long)hi);
        auto res = resolve(pack);                                                   // :835
        if (res.second) {                                                           // :835-837: rows stay as they are now
            invalidate_device();
            return res.second;
        }
        if (res.first.size() >= impl_columns.size()) impl_rows[dest++] = std::move(res.first);   // :842-845
        const size_t stop = k + 1 < g->ngroups ? (size_t)g->lower[k + 1] : n - 1;   // :848-859
        for (size_t i = hi; i < stop; i++, dest++)
            if (dest != i) impl_rows[dest] = impl_rows[i];   // a copy, as the reference's slice assignment: the source slot keeps its row
    }
    impl_rows.resize(dest);                                                         // :862-864
    invalidate_device();   // the device twin is rebuilt from the surviving rows on next use
    return Error();
}

namespace detail {
inline void put_u64(std::ostream& o, uint64_t v) {
    char b[8];
    for (int i = 0; i < 8; i++) b[i] = (char)(v >> (8 * i));
    o.write(b, 8);
}
inline void put_str(std::ostream& o, const std::string& s) {
    put_u64(o, s.size());
    o.write(s.data(), (std::streamsize)s.size());
}
inline bool get_u64(std::istream& in, uint64_t* v) {
    unsigned char b[8];
    if (!in.read(reinterpret_cast<char*>(b), 8)) return false;
    *v = 0;
    for (int i = 0; i < 8; i++) *v |= (uint64_t)b[i] << (8 * i);
    return true;
}
inline bool get_str(std::istream& in, std::string* s, uint64_t limit) {
    uint64_t n;
    if (!get_u64(in, &n) || n > limit) return false;
    s->resize((size_t)n);
    return n == 0 || (bool)in.read(&(*s)[0], (std::streamsize)n);
}
constexpr char kIndexMagic[8] = {'C', 'S', 'V', 'P', 'I', 'D', 'X', '1'};
}  // namespace detail

inline Error Index::WriteTo(const std::string& fileName) const {
    std::ofstream f(fileName, std::ios::binary | std::ios::trunc);
    if (!f) return Error("open " + fileName + ": cannot create file");
    f.write(detail::kIndexMagic, 8);
    detail::put_u64(f, impl_columns.size());                                        // enc.Encode(columns) :674
    for (const auto& c : impl_columns) detail::put_str(f, c);
    detail::put_u64(f, impl_rows.size());                                           // enc.Encode(rows) :675
    for (const Row& r : impl_rows) {
        detail::put_u64(f, r.size());
        for (const auto& kv : r) {
            detail::put_str(f, kv.first);
            detail::put_str(f, kv.second);
        }
    }
    f.close();
    if (!f) {                                                                       // :663-671: no partial files
        std::remove(fileName.c_str());
        return Error("write " + fileName + ": short write");
    }
    return Error();
}

// LoadIndex (:682-705).  Like the reference it trusts the file: rows are taken to be sorted on `columns`.
inline std::pair<std::shared_ptr<Index>, Error> LoadIndex(const std::string& fileName) {
    std::ifstream f(fileName, std::ios::binary);
    if (!f) return {nullptr, Error("open " + fileName + ": no such file or directory")};
    f.seekg(0, std::ios::end);
    const uint64_t size = (uint64_t)f.tellg();
    f.seekg(0);
    const Error bad(fileName + ": not a csvplus index file");
    char magic[8];
    if (!f.read(magic, 8) || std::memcmp(magic, detail::kIndexMagic, 8) != 0) return {nullptr, bad};
    auto index = std::make_shared<Index>();
    uint64_t ncols, nrows;
    if (!detail::get_u64(f, &ncols) || ncols > size) return {nullptr, bad};
    index->impl_columns.resize((size_t)ncols);
    for (auto& c : index->impl_columns)
        if (!detail::get_str(f, &c, size)) return {nullptr, bad};
    if (!detail::get_u64(f, &nrows) || nrows > size) return {nullptr, bad};
    index->impl_rows.resize((size_t)nrows);
    for (Row& r : index->impl_rows) {
        uint64_t nf;
        if (!detail::get_u64(f, &nf) || nf > size) return {nullptr, bad};
        for (uint64_t i = 0; i < nf; i++) {
            std::string k, v;
            if (!detail::get_str(f, &k, size) || !detail::get_str(f, &v, size)) return {nullptr, bad};
            r.emplace(std::move(k), std::move(v));
        }
    }
    return {index, Error()};
}

}  // namespace csvplus

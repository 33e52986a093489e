// test_host.cpp — the reference's own hot-path tests (csvplus_test.go) restated against the
// C++ host facade (csvplus_amd/host/csvplus.hpp), which runs the sort / unique check / probe on
// the GPU through the C ABI.  Fixtures have the reference's shape (csvplus_test.go:1207-1333)
// with a seeded PRNG (the reference's math/rand is unseeded).  Run by tests/test_host_cpp.py
// under `-m gpu`.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <random>
#include <sstream>

#include "csvplus.hpp"

using namespace csvplus;

static int g_failed = 0;
#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::printf("  CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
            g_failed++;                                                                \
            return;                                                                    \
        }                                                                              \
    } while (0)

// ---- generated test data (csvplus_test.go:1188-1333) ----------------------------------------------
static const char* peopleNames[] = {"Amelia", "Olivia", "Emily", "Ava", "Isla", "Oliver", "Jack", "Harry", "Jacob", "Charlie"};
static const char* peopleSurnames[] = {"Smith", "Jones", "Taylor", "Williams", "Brown", "Davies",
                                       "Evans", "Wilson", "Thomas", "Roberts", "Johnson", "Lewis"};
static const int kNames = 10, kSurnames = 12, numOrders = 10000;
struct StockItem { const char* name; double price; };
static const StockItem stockItems[] = {{"banana", 0.01}, {"apple", 0.02}, {"orange", 0.03}, {"pea", 0.04},
                                       {"tomato", 0.05}, {"potato", 0.06}, {"cucumber", 0.07}, {"iPhone", 0.08}};
static const int kStock = 8;

struct PersonData { std::string Name, Surname; int Born; };
struct OrderData { int custID, prodID, qty; };
static std::vector<PersonData> peopleData;
static std::vector<OrderData> ordersData;
static std::vector<Row> peopleRows, ordersRows, stockRows;

static void makeFixtures() {
    std::mt19937_64 rng(20250523);
    for (int i = 0; i < kNames; i++)
        for (int j = 0; j < kSurnames; j++) {
            int id = i * kSurnames + j;
            PersonData p{peopleNames[i], peopleSurnames[j], 1916 + (int)(rng() % 90)};
            peopleData.push_back(p);
            peopleRows.push_back(Row{{"id", std::to_string(id)}, {"name", p.Name}, {"surname", p.Surname},
                                     {"born", std::to_string(p.Born)}});
        }
    for (int i = 0; i < kStock; i++) {
        char price[16];
        std::snprintf(price, sizeof price, "%.2f", stockItems[i].price);
        stockRows.push_back(Row{{"prod_id", std::to_string(i)}, {"product", stockItems[i].name}, {"price", price}});
    }
    for (int i = 0; i < numOrders; i++) {
        OrderData o{(int)(rng() % (kNames * kSurnames)), (int)(rng() % kStock), (int)(rng() % 100) + 1};
        ordersData.push_back(o);
        ordersRows.push_back(Row{{"order_id", std::to_string(i)}, {"cust_id", std::to_string(o.custID)},
                                 {"prod_id", std::to_string(o.prodID)}, {"qty", std::to_string(o.qty)},
                                 {"ts", "2025-05-23T00:00:00Z"}});
    }
}

// ---- the lazy combinators the reference's tests chain around the joins (out of scope for the
// GPU path, so they live here as plain host helpers; csvplus.go:258-374, :492-525) ----------------
static DataSource SelectColumns(DataSource src, std::vector<std::string> cols) {
    return DataSource([src, cols](const RowFunc& fn) {
        return src([&](Row row) {
            Row r;
            for (auto& c : cols) {
                auto it = row.find(c);
                if (it == row.end()) return Error("missing column " + quote(c));
                r[c] = it->second;
            }
            return fn(std::move(r));
        });
    });
}
static DataSource Filter(DataSource src, std::function<bool(const Row&)> pred) {
    return DataSource([src, pred](const RowFunc& fn) {
        return src([&](Row row) { return pred(row) ? fn(std::move(row)) : Error(); });
    });
}
static DataSource Map(DataSource src, std::function<Row(Row)> mf) {
    return DataSource([src, mf](const RowFunc& fn) { return src([&](Row row) { return fn(mf(std::move(row))); }); });
}
static DataSource Top(DataSource src, uint64_t n) {   // csvplus.go:313-325
    return DataSource([src, n](const RowFunc& fn) {
        uint64_t counter = 0;
        return src([&](Row row) {
            if (counter++ < n) return fn(std::move(row));
            return io_EOF;
        });
    });
}
static int atoi_s(const std::string& s) { return std::atoi(s.c_str()); }
static bool hasSuffix(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// ---- TestIndexImpl (csvplus_test.go:198-246) ------------------------------------------------------------
static void TestIndexImpl() {
    std::vector<Row> rows = {
        {{"x", "1"}, {"y", "2"}, {"z", "3"}, {"junk", "zzz"}}, {{"x", "5"}, {"y", "6"}, {"z", "8"}, {"junk", "nnn"}},
        {{"x", "0"}, {"y", "5"}, {"z", "3"}, {"junk", "xxx"}}, {{"x", "8"}, {"y", "9"}, {"z", "1"}, {"junk", "aaa"}},
        {{"x", "7"}, {"y", "4"}, {"z", "0"}, {"junk", "bbb"}}, {{"x", "5"}, {"y", "6"}, {"z", "9"}, {"junk", "iii"}},
        {{"x", "2"}, {"y", "6"}, {"z", "7"}, {"junk", "mmm"}},
    };
    auto [index, err] = TakeRows(rows).IndexOn({"x", "y", "z"});
    CHECK(!err);
    auto r = index->find({"1", "2", "3"});
    CHECK(r.second - r.first == 1 && index->rows()[r.first].at("junk") == "zzz");
    r = index->find({"5", "6", "8"});
    CHECK(r.second - r.first == 1 && index->rows()[r.first].at("junk") == "nnn");
    r = index->find({"5", "6"});
    CHECK(r.second - r.first == 2);
    for (size_t i = r.first; i < r.second; i++) CHECK(index->rows()[i].at("x") == "5" && index->rows()[i].at("y") == "6");
    const char* want[] = {"xxx", "zzz", "mmm", "nnn", "iii", "bbb", "aaa"};
    for (int i = 0; i < 7; i++) CHECK(index->rows()[i].at("junk") == want[i]);
}

// ---- TestSimpleUniqueJoin (csvplus_test.go:368-452) ---------------------------------------------------------
static void TestSimpleUniqueJoin() {
    auto people = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"});
    auto orders = SelectColumns(TakeRows(ordersRows), {"order_id", "cust_id", "qty"});
    auto [idIndex, err] = people.UniqueIndexOn({"id"});
    CHECK(!err);
    std::vector<int> qtyMap(peopleData.size(), 0);
    uint64_t seen = 0;
    err = orders.Join(idIndex, {"cust_id"})([&](Row row) -> Error {
        int id = atoi_s(row.at("id")), orderID = atoi_s(row.at("order_id")), custID = atoi_s(row.at("cust_id")),
            qty = atoi_s(row.at("qty"));
        if (id >= (int)peopleData.size()) return Error("Invalid id");
        if (peopleData[id].Name != row.at("name") || peopleData[id].Surname != row.at("surname"))
            return Error("Invalid parameters associated with id");
        if (id != custID) return Error("id != cust_id");
        if (orderID != (int)seen) return Error("orders out of stream order");   // emission = stream order
        if (ordersData[orderID].custID != custID || ordersData[orderID].qty != qty) return Error("bad order data");
        if (row.size() != 6) return Error("Invalid number of columns");
        qtyMap[id] += qty;
        seen++;
        return Error();
    });
    if (err) std::printf("  Join failed: %s\n", err.message().c_str());
    CHECK(!err);
    CHECK(seen == (uint64_t)numOrders);
    std::vector<int> origMap(peopleData.size(), 0);
    for (auto& d : ordersData) origMap[d.custID] += d.qty;
    CHECK(qtyMap == origMap);
}

// ---- TestSorted (csvplus_test.go:454-514) ------------------------------------------------------------------------
static void TestSorted() {
    auto people = TakeRows(peopleRows);
    auto [index, err] = people.UniqueIndexOn({"name", "surname"});
    CHECK(!err);
    for (int i = 0; i < kSurnames; i++) CHECK(index->rows()[i].at("name") == "Amelia");
    for (int i = kSurnames; i < 2 * kSurnames; i++) CHECK(index->rows()[i].at("name") == "Ava");
    int n = 0;
    err = Top(Take(index), kSurnames)([&](Row row) { n++; return row.at("name") == "Amelia" ? Error() : Error("Unexpected name"); });
    CHECK(!err && n == kSurnames);
    std::tie(index, err) = people.UniqueIndexOn({"surname", "name"});
    CHECK(!err);
    for (int i = kNames; i < 2 * kNames; i++) CHECK(index->rows()[i].at("surname") == "Davies");
}

// ---- TestSimpleTotals (csvplus_test.go:516-571): natural join ---------------------------------------------------------
static void TestSimpleTotals() {
    auto orders = SelectColumns(TakeRows(ordersRows), {"cust_id", "prod_id", "qty"});
    auto products = SelectColumns(TakeRows(stockRows), {"prod_id", "price"});
    auto [prodIndex, err] = products.UniqueIndexOn({"prod_id"});
    CHECK(!err);
    std::vector<double> totals(peopleData.size(), 0), orig(peopleData.size(), 0);
    err = orders.Join(prodIndex)([&](Row row) {
        int id = atoi_s(row.at("cust_id"));
        totals[id] = std::atof(row.at("price").c_str()) * atoi_s(row.at("qty"));
        return Error();
    });
    CHECK(!err);
    for (auto& o : ordersData) orig[o.custID] = stockItems[o.prodID].price * o.qty;
    for (size_t i = 0; i < totals.size(); i++) CHECK(std::fabs((totals[i] - orig[i]) / totals[i]) <= 1e-6);
}

// ---- TestLongChain (csvplus_test.go:248-366) -----------------------------------------------------------------------------
static std::vector<Row> nestedJoinOnHost(const std::vector<Row>& stream, const Index& ia, const std::vector<std::string>& ka,
                                         const Index& ib, const std::vector<std::string>& kb, Error* err);
static void TestLongChain() {
    auto [orders, err] = SelectColumns(TakeRows(ordersRows), {"order_id", "cust_id", "prod_id", "qty", "ts"}).IndexOn({"cust_id"});
    CHECK(!err);
    auto [products, err2] = SelectColumns(TakeRows(stockRows), {"prod_id", "product", "price"}).UniqueIndexOn({"prod_id"});
    CHECK(!err2);
    auto people = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname", "born"});
    int n = 0;
    // the reference's pipeline, method for method (csvplus_test.go:270-293).  Round 5: the two Joins and the DropColumns between
    // them are ONE chain: prod_id — the key of the second Join — is a column of the ORDERS index rows, which the device reads
    // from the row the first Join matched (cph_chain_step.source): one fused device call for the batch, no host round trip per step.
    auto chain = Top(Filter(Map(Filter(people, [](const Row& r) { return atoi_s(r.at("born")) > 1970; })
                                    .SelectColumns({"id", "name", "surname"})
                                    .Join(orders, {"id"})
                                    .DropColumns({"ts", "order_id", "cust_id"})
                                    .Join(products)
                                    .DropColumns({"prod_id"}),
                                [](Row row) { if (row["name"] == "Amelia") row["name"] = "Julia"; return row; }),
                            [](const Row& r) { return r.at("surname") == "Smith"; }),
                     10)
                     .DropColumns({"id"});
    const uint64_t calls_before = DataSource::fused_calls();
    Error e = chain([&](Row row) -> Error {
        if (++n > 10) return Error("Too many rows");
        if (row.at("surname") != "Smith") return Error("Surname \"Smith\" not found");
        if (row.at("name") == "Amelia") return Error("Name \"Amelia\" found");
        if (!SelectExisting(row, {"born", "ts", "order_id", "prod_id", "cust_id"}).empty()) return Error("Some deleted fields are still there");
        if (row.size() != 5) return Error("Unexpected number of columns: " + std::to_string(row.size()));
        return Error();
    });
    if (e) std::printf("  chain: %s\n", e.message().c_str());
    CHECK(!e);
    CHECK(n == 10);
    CHECK(DataSource::fused_calls() == calls_before + 1);   // the whole (one-batch) pipeline was ONE cph_join_chain_ex call
    {
        // all rows of the same chain against the nested joins run on the host (duplicates on the first build side: ~83 orders per person)
        auto people3 = TakeRows(peopleRows).SelectColumns({"id", "name", "surname"});
        Error he;
        std::vector<Row> stream;
        Error se = people3([&](Row r) { stream.push_back(std::move(r)); return Error(); });
        CHECK(!se);
        std::vector<Row> want = nestedJoinOnHost(stream, *orders, {"id"}, *products, {"prod_id"}, &he);
        for (size_t batch : {(size_t)1, (size_t)50, (size_t)8192}) {
            Gpu::Default().join_batch_rows = batch;
            const uint64_t c0 = DataSource::fused_calls();
            auto [got, ge] = people3.Join(orders, {"id"}).Join(products).ToRows();
            CHECK(!he && !ge && got.size() == (size_t)numOrders && got == want);
            CHECK(DataSource::fused_calls() == c0 + (stream.size() + batch - 1) / batch);
        }
        Gpu::Default().join_batch_rows = 8192;
        // a DropColumns that removes the later key: the reference fails with `missing column "prod_id"` on the first joined row
        auto [none, me] = people3.Join(orders, {"id"}).DropColumns({"prod_id"}).Join(products).ToRows();
        CHECK(me && me.message() == "missing column \"prod_id\"" && none.empty());
    }
    // the indices are unchanged afterwards (:325-365)
    n = 0;
    e = Take(orders)([&](Row row) { n++; return row.size() == 5 ? Error() : Error("bad order row"); });
    CHECK(!e && n == numOrders);
    n = 0;
    e = Take(products)([&](Row) { n++; return Error(); });
    CHECK(!e && n == kStock);
}

// ---- TestMultiIndex (csvplus_test.go:573-649) -------------------------------------------------------------------------------
static void TestMultiIndex() {
    auto source = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"});
    auto [index, err] = source.UniqueIndexOn({"name", "surname"});
    CHECK(!err);
    auto neverCalled = [](Row) { return Error("This must never be called"); };
    CHECK(!index->Find({"xxx"})(neverCalled));
    int cnt = 0;
    CHECK(!index->Find({"Amelia"})([&](Row row) { cnt++; return row.at("name") == "Amelia" ? Error() : Error("bad name"); }));
    CHECK(cnt == kSurnames);
    for (int i = 0; i < kNames; i++) {   // self-join on each sub-index (:601-624)
        std::map<std::string, int> surnames;
        auto s = source.Join(index->SubIndex({peopleNames[i]}));
        CHECK(!s([&](Row row) { surnames[row.at("surname")]++; return Error(); }));
        CHECK((int)surnames.size() == kSurnames);
        for (int j = 0; j < kSurnames; j++) CHECK(surnames[peopleSurnames[j]] == kNames);
    }
    for (auto& p : peopleData) {
        int count = 0;
        CHECK(!index->Find({p.Name, p.Surname})([&](Row) { count++; return Error(); }));
        CHECK(count == 1);
    }
    CHECK(!index->Find({"Jack", "xxx"})(neverCalled));
}

// ---- TestExcept (csvplus_test.go:651-693) --------------------------------------------------------------------------------------
static void TestExcept() {
    const std::string name = "Emily";
    auto [people, err] = Filter(SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"}),
                                [&](const Row& r) { return r.at("name") == name; }).IndexOn({"id"});
    CHECK(!err);
    int n = 0, m = 0;
    Error e = SelectColumns(TakeRows(ordersRows), {"cust_id", "prod_id", "qty"}).Except(people, {"cust_id"})([&](Row row) {
        if (peopleData[atoi_s(row.at("cust_id"))].Name == name) return Error("Cust. id somehow got through");
        n++;
        return Error();
    });
    CHECK(!e);
    for (auto& o : ordersData) if (peopleData[o.custID].Name != name) m++;
    CHECK(n == m);
}

// ---- TestErrors, index-related parts (csvplus_test.go:825-883) --------------------------------------------------------------------
static void TestErrors() {
    auto source = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"});
    auto [ix, err] = source.IndexOn({"name", "xxx"});
    CHECK(ix == nullptr && err && hasSuffix(err.message(), "missing column \"xxx\" while creating an index"));
    CHECK(err.is_data_source_error() && err.line() == 0);   // iterate wraps with the 0-based row (:242-245)
    std::tie(ix, err) = source.UniqueIndexOn({"name"});
    CHECK(ix == nullptr && err && err.message().find("duplicate value while creating unique index:") != std::string::npos);
    CHECK(err.message() == "duplicate value while creating unique index: { \"name\" : \"Amelia\" }");
    bool panicked = false;
    try { source.IndexOn({}); } catch (const Panic&) { panicked = true; }
    CHECK(panicked);
    panicked = false;
    try { source.IndexOn({"id", "name", "id"}); } catch (const Panic&) { panicked = true; }
    CHECK(panicked);
    std::tie(ix, err) = source.IndexOn({"id"});
    CHECK(!err);
    panicked = false;
    try { ix->SubIndex({"aaa", "bbb"}); } catch (const Panic&) { panicked = true; }
    CHECK(panicked);
    panicked = false;
    try { source.Join(ix, {"id", "name"}); } catch (const Panic&) { panicked = true; }   // :548-550
    CHECK(panicked);
    // missing join column: rows before the bad one are delivered, then the error surfaces (:556, :145)
    std::vector<Row> rows = {{{"id", "3"}}, {{"id", "4"}}, {{"nope", "1"}}, {{"id", "5"}}};
    int delivered = 0;
    Error e = TakeRows(rows).Join(ix)([&](Row) { delivered++; return Error(); });
    CHECK(e && hasSuffix(e.message(), "missing column \"id\"") && delivered == 2);
    CHECK(e.is_data_source_error() && e.line() == 2);
}

// ---- TestResolver (csvplus_test.go:695-752) + the resolver part of TestErrors (:843-863) ----------------------------------------
static void TestResolver() {
    std::vector<Row> source;
    CHECK(!SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"})([&](Row r) { source.push_back(std::move(r)); return Error(); }));
    std::mt19937_64 rng(695);
    for (int i = 0; i < 25; i++) {
        std::vector<Row> src = source;
        const Row dup = src[rng() % src.size()];
        const int n = (int)(rng() % 100) + 1;
        for (int j = 0; j < n; j++) {
            size_t k = rng() % src.size();
            src.push_back(dup);
            std::swap(src[k], src.back());
        }
        auto [index, err] = TakeRows(src).IndexOn({"name", "surname"});
        CHECK(!err);
        int nc = 0;
        Error e = index->ResolveDuplicates([&](const std::vector<Row>& rows) -> std::pair<Row, Error> {
            if (++nc != 1) return {Row{}, Error("Unexpected second call to the resolution function")};
            if ((int)rows.size() != n + 1) return {Row{}, Error("Unexpected number of duplicates")};
            for (const Row& r : rows)
                if (r != dup) return {Row{}, Error("Unexpected duplicate")};
            return {rows[0], Error()};
        });
        CHECK(!e && nc == 1);
        // the surviving rows: one per key, in key order; the reference's tail rule (csvplus.go:851-859) costs the
        // final row unless the resolved pack was the last one
        const auto& rows = index->rows();
        const bool dupIsLast = dup.at("name") == "Olivia" && dup.at("surname") == "Wilson";   // greatest (name, surname)
        CHECK(rows.size() == source.size() - (dupIsLast ? 0 : 1));
        for (size_t k = 1; k < rows.size(); k++)
            CHECK(std::make_pair(rows[k - 1].at("name"), rows[k - 1].at("surname")) < std::make_pair(rows[k].at("name"), rows[k].at("surname")));
        // and the deduplicated index keeps working as a join target
        size_t hits = 0;
        CHECK(!TakeRows(source).Join(index)([&](Row) { hits++; return Error(); }));
        CHECK(hits == rows.size());
    }
    // every key duplicated (TestErrors :843-863): 10 names x 12 surnames -> 10 rows, the last pack reaches the end
    auto [byName, err] = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"}).IndexOn({"name"});
    CHECK(!err);
    Error e = byName->ResolveDuplicates([&](const std::vector<Row>& rows) -> std::pair<Row, Error> {
        if ((int)rows.size() != kSurnames) return {Row{}, Error("Unexpected number of duplicate rows")};
        return {rows[0], Error()};
    });
    CHECK(!e && (int)byName->rows().size() == kNames);
    // hand-derived cases of csvplus.go:810-867: [A,A,B] -> [A]; [A,A,B,C] -> [A,B]; empty row drops the pack
    auto keysAfter = [&](const std::string& keys, bool drop) -> std::string {
        std::vector<Row> rows;
        for (size_t k = 0; k < keys.size(); k++) rows.push_back(Row{{"k", std::string(1, keys[k])}, {"id", std::to_string(k)}});
        auto [ix, er] = TakeRows(rows).IndexOn({"k"});
        if (er) return "IndexOn failed";
        if (ix->ResolveDuplicates([&](const std::vector<Row>& pack) -> std::pair<Row, Error> {
                return {drop ? Row{} : pack[0], Error()};
            }))
            return "ResolveDuplicates failed";
        std::string out;
        for (const Row& r : ix->rows()) out += r.at("k");
        return out;
    };
    CHECK(keysAfter("AAB", false) == "A");
    CHECK(keysAfter("BAAC", false) == "AB");
    CHECK(keysAfter("ABB", false) == "AB");
    CHECK(keysAfter("CAB", false) == "ABC");
    CHECK(keysAfter("AABCC", true) == "B");
    // a resolver error is returned as is and stops the pass (:835-837)
    auto [ix3, er3] = TakeRows(std::vector<Row>{{{"k", "x"}}, {{"k", "x"}}, {{"k", "y"}}, {{"k", "y"}}}).IndexOn({"k"});
    CHECK(!er3);
    int calls = 0;
    e = ix3->ResolveDuplicates([&](const std::vector<Row>&) -> std::pair<Row, Error> { calls++; return {Row{}, Error("stop")}; });
    CHECK(e && e.message() == "stop" && calls == 1 && ix3->rows().size() == 4);
}

// ---- TestIndexStore (csvplus_test.go:959-1013) ----------------------------------------------------------------------------------
static void TestIndexStore() {
    auto [index, err] = SelectColumns(TakeRows(peopleRows), {"id", "name", "surname"}).IndexOn({"id"});
    CHECK(!err);
    const std::string path = "/tmp/csvplus_amd_test_index.bin";
    CHECK(!index->WriteTo(path));
    auto [index2, err2] = LoadIndex(path);
    CHECK(!err2 && index2);
    CHECK(index2->columns() == index->columns());
    CHECK(index2->rows() == index->rows());
    // the loaded index is usable: Find goes through the rebuilt device twin
    size_t found = 0;
    CHECK(!index2->Find({"17"})([&](Row r) { found += r.at("id") == "17"; return Error(); }));
    CHECK(found == 1);
    std::remove(path.c_str());
    auto [none, err3] = LoadIndex(path);
    CHECK(none == nullptr && err3);
    CHECK(index->WriteTo("/nonexistent-dir/x.bin"));
}

// ---- laziness across the batched boundary ------------------------------------------------------------------------
static void TestBatchingSemantics() {
    auto [ix, err] = SelectColumns(TakeRows(peopleRows), {"id", "name"}).UniqueIndexOn({"id"});
    CHECK(!err);
    auto orders = SelectColumns(TakeRows(ordersRows), {"order_id", "cust_id"});
    std::vector<std::string> ref;
    for (size_t batch : {(size_t)1, (size_t)7, (size_t)8192, (size_t)100000}) {
        Gpu::Default().join_batch_rows = batch;
        std::vector<std::string> got;
        Error e = orders.Join(ix, {"cust_id"})([&](Row row) { got.push_back(row.at("order_id") + ":" + row.at("name")); return Error(); });
        CHECK(!e && got.size() == (size_t)numOrders);
        if (ref.empty()) ref = got; else CHECK(got == ref);
        // early stop: io.EOF after 5 rows ends the whole pipeline cleanly
        int k = 0;
        e = orders.Join(ix, {"cust_id"})([&](Row) { return ++k == 5 ? io_EOF : Error(); });
        CHECK(!e && k == 5);
        // a callback error aborts and is reported through the source's DataSourceError wrapper
        k = 0;
        e = orders.Join(ix, {"cust_id"})([&](Row) { return ++k == 3 ? Error("boom") : Error(); });
        CHECK(e && e.message().find("boom") != std::string::npos && k == 3);
    }
    Gpu::Default().join_batch_rows = 8192;
    // stream value wins on a column-name collision (mergeRows :578-580)
    std::vector<Row> stream = {{{"id", "7"}, {"name", "STREAM"}}};
    Error e = TakeRows(stream).Join(ix)([&](Row row) { return row.at("name") == "STREAM" ? Error() : Error("index value won"); });
    CHECK(!e);
}


// ---- chained Joins through the facade (round 4): src.Join(a).Join(b) is ONE fused device call per batch when every stream
// row carries all key columns, the steps one after the other otherwise; either way the reference's nested semantics
// (csvplus.go:545-569 nested, mergeRows :571-583): precedence stream > first index > second index on a shared column
// name, emission order by stream row, then position in the first index, then in the second; early stop and errors. ----------
static std::vector<Row> nestedJoinOnHost(const std::vector<Row>& stream, const Index& ia, const std::vector<std::string>& ka,
                                         const Index& ib, const std::vector<std::string>& kb, Error* err) {
    // the reference's closures, literally: for every stream row, for every index row with equal key columns (in index order) ...
    auto matches = [](const Index& ix, const std::vector<std::string>& cols, const Row& row, std::vector<const Row*>* out, Error* e) {
        out->clear();
        std::vector<const std::string*> vals;
        *e = SelectValues(row, cols, &vals);
        if (*e) return;
        for (const Row& r : ix.rows()) {
            bool eq = true;
            for (size_t c = 0; c < cols.size() && eq; c++) eq = r.at(ix.columns()[c]) == *vals[c];
            if (eq) out->push_back(&r);
        }
    };
    std::vector<Row> out;
    std::vector<const Row*> ma, mb;
    for (const Row& s : stream) {
        matches(ia, ka, s, &ma, err);
        if (*err) return out;
        for (const Row* a : ma) {
            Row r1 = mergeRows(*a, s);
            matches(ib, kb, r1, &mb, err);
            if (*err) return out;
            for (const Row* b : mb) out.push_back(mergeRows(*b, r1));
        }
    }
    return out;
}

static void TestChainPrecedence() {
    std::mt19937_64 rng(4);
    // customers and products share the column names "name" and "tag"; some stream rows carry a "name" of their own
    std::vector<Row> cust, prod, dupcust;
    for (int i = 0; i < 300; i++)
        cust.push_back(Row{{"id", std::to_string(i)}, {"name", "cust-" + std::to_string(i)}, {"tag", "C"}, {"fav_prod", std::to_string(i % 40)}});
    for (int i = 0; i < 40; i++)
        prod.push_back(Row{{"prod_id", std::to_string(i)}, {"name", "prod-" + std::to_string(i)}, {"tag", "P"}, {"price", std::to_string(i) + ".99"}});
    for (int i = 0; i < 900; i++)   // three rows per customer id: duplicates in the first index
        dupcust.push_back(Row{{"id", std::to_string(i % 300)}, {"name", "dup-" + std::to_string(i)}, {"tag", "D"}});
    auto [ic, e1] = TakeRows(cust).UniqueIndexOn({"id"});
    auto [ip, e2] = TakeRows(prod).UniqueIndexOn({"prod_id"});
    auto [id, e3] = TakeRows(dupcust).IndexOn({"id"});
    CHECK(!e1 && !e2 && !e3);
    std::vector<Row> stream;
    for (int i = 0; i < 5000; i++) {
        Row r{{"order_id", std::to_string(i)}, {"cust_id", std::to_string(rng() % 330)}, {"prod_id", std::to_string(rng() % 44)}};
        if (i % 3 == 0) r["name"] = "stream-" + std::to_string(i);
        stream.push_back(r);
    }
    auto same = [](const std::vector<Row>& a, const std::vector<Row>& b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); i++)
            if (a[i] != b[i]) return false;
        return true;
    };
    for (size_t batch : {(size_t)1, (size_t)7, (size_t)8192}) {
        Gpu::Default().join_batch_rows = batch;
        // (1) fused: both keys from the stream row
        Error he;
        std::vector<Row> want = nestedJoinOnHost(stream, *ic, {"cust_id"}, *ip, {"prod_id"}, &he);
        CHECK(!he && !want.empty());
        auto [got, ge] = TakeRows(stream).Join(ic, {"cust_id"}).Join(ip, {"prod_id"}).ToRows();
        CHECK(!ge);
        CHECK(same(got, want));
        for (const Row& r : got) {   // stream > customers > products
            const bool own = atoi_s(r.at("order_id")) % 3 == 0;
            CHECK(r.at("tag") == "C");
            CHECK(own ? r.at("name").rfind("stream-", 0) == 0 : r.at("name").rfind("cust-", 0) == 0);
        }
        // (2) duplicates in the first index (three matches per stream row, in index order), fused
        want = nestedJoinOnHost(stream, *id, {"cust_id"}, *ip, {"prod_id"}, &he);
        auto [got2, ge2] = TakeRows(stream).Join(id, {"cust_id"}).Join(ip, {"prod_id"}).ToRows();
        CHECK(!he && !ge2 && same(got2, want) && got2.size() > got.size());
        // (3) the second key comes from the FIRST INDEX's row ("fav_prod" is a customers column): fused, the device gathers it
        want = nestedJoinOnHost(stream, *ic, {"cust_id"}, *ip, {"fav_prod"}, &he);
        const uint64_t f3 = DataSource::fused_calls();
        auto [got3, ge3] = TakeRows(stream).Join(ic, {"cust_id"}).Join(ip, {"fav_prod"}).ToRows();
        CHECK(!he && !ge3 && same(got3, want) && !got3.empty());
        CHECK(DataSource::fused_calls() == f3 + (stream.size() + batch - 1) / batch);
        // (4) mixed: every fifth stream row brings its own "fav_prod" (which then wins over the customer's)
        std::vector<Row> mixed = stream;
        for (size_t i = 0; i < mixed.size(); i += 5) mixed[i]["fav_prod"] = std::to_string(i % 44);
        want = nestedJoinOnHost(mixed, *ic, {"cust_id"}, *ip, {"fav_prod"}, &he);
        auto [got4, ge4] = TakeRows(mixed).Join(ic, {"cust_id"}).Join(ip, {"fav_prod"}).ToRows();
        CHECK(!he && !ge4 && same(got4, want));
        // (5) three steps, the last one natural on the first index's own key column
        auto three = TakeRows(stream).Join(ic, {"cust_id"}).Join(ip, {"prod_id"}).Join(id, {"cust_id"});
        size_t n3 = 0;
        Error e5 = three([&](Row r) { n3++; return r.at("tag") == "C" ? Error() : Error("precedence"); });
        CHECK(!e5 && n3 == 3 * got.size());
        // (5b) six Joins in a row: one device call per batch (CPH_MAX_CHAIN = 8 since round 5; the general chain runs on the device),
        //      same rows as three nested pairs on the host
        //      (the whole stream in one batch; the first 70 rows in batches of 7 and of 1: a general-chain call is ~20 launches)
        {
            const std::vector<Row> part(stream.begin(), stream.begin() + (batch == 8192 ? (long)stream.size() : 70));
            std::vector<Row> w2 = nestedJoinOnHost(part, *ic, {"cust_id"}, *ip, {"prod_id"}, &he);
            std::vector<Row> w4 = nestedJoinOnHost(w2, *id, {"cust_id"}, *ip, {"fav_prod"}, &he);
            std::vector<Row> w6 = nestedJoinOnHost(w4, *ic, {"id"}, *ip, {"prod_id"}, &he);
            CHECK(!he && !w6.empty());
            const uint64_t f6 = DataSource::fused_calls();
            auto [got6, ge6] = TakeRows(part).Join(ic, {"cust_id"}).Join(ip, {"prod_id"}).Join(id, {"cust_id"}).Join(ip, {"fav_prod"})
                                   .Join(ic, {"id"}).Join(ip, {"prod_id"}).ToRows();
            CHECK(!ge6 && same(got6, w6));
            CHECK(DataSource::fused_calls() == f6 + (part.size() + batch - 1) / batch);
        }
        // (6) early stop through both Joins: io.EOF ends the pipeline cleanly after exactly 5 rows; an error is reported
        int k = 0;
        Error e6 = TakeRows(stream).Join(ic, {"cust_id"}).Join(ip, {"prod_id"})([&](Row) { return ++k == 5 ? io_EOF : Error(); });
        CHECK(!e6 && k == 5);
        k = 0;
        e6 = TakeRows(stream).Join(id, {"cust_id"}).Join(ip, {"prod_id"})([&](Row) { return ++k == 4 ? Error("boom") : Error(); });
        CHECK(e6 && e6.message().find("boom") != std::string::npos && k == 4);
        // (7) a row that reaches the second Join without its key column: the rows in front of it are delivered, then the
        //     error surfaces (csvplus.go:556, :145); a row that does NOT reach the second Join (no customer) raises nothing
        std::vector<Row> holes;
        for (int i = 0; i < 40; i++) holes.push_back(Row{{"order_id", std::to_string(i)}, {"cust_id", std::to_string(i)}, {"prod_id", std::to_string(i % 40)}});
        holes[9].erase("prod_id");
        holes[9]["cust_id"] = "99999";          // joins nothing: never seen by the second Join
        holes[20].erase("prod_id");             // joins customer 20: the second Join misses "prod_id"
        want = nestedJoinOnHost(holes, *ic, {"cust_id"}, *ip, {"prod_id"}, &he);
        CHECK(he && want.size() == 19);
        std::vector<Row> got7;
        Error e7 = TakeRows(holes).Join(ic, {"cust_id"}).Join(ip, {"prod_id"})([&](Row r) { got7.push_back(std::move(r)); return Error(); });
        CHECK(e7 && e7.message().find("missing column \"prod_id\"") != std::string::npos);
        CHECK(same(got7, want));
    }
    Gpu::Default().join_batch_rows = 8192;
}

int main() {
    makeFixtures();
    struct T { const char* name; void (*fn)(); };
    const T tests[] = {{"TestIndexImpl", TestIndexImpl}, {"TestSimpleUniqueJoin", TestSimpleUniqueJoin},
                       {"TestSorted", TestSorted}, {"TestSimpleTotals", TestSimpleTotals},
                       {"TestLongChain", TestLongChain}, {"TestMultiIndex", TestMultiIndex},
                       {"TestExcept", TestExcept}, {"TestErrors", TestErrors},
                       {"TestResolver", TestResolver}, {"TestIndexStore", TestIndexStore},
                       {"TestBatchingSemantics", TestBatchingSemantics}, {"TestChainPrecedence", TestChainPrecedence}};
    int bad = 0;
    for (auto& t : tests) {
        int before = g_failed;
        try {
            t.fn();
        } catch (const std::exception& e) {
            std::printf("  exception: %s\n", e.what());
            g_failed++;
        }
        std::printf("%s %s\n", g_failed == before ? "PASS" : "FAIL", t.name);
        if (g_failed != before) bad++;
    }
    std::printf("%d of %zu host tests failed\n", bad, sizeof tests / sizeof tests[0]);
    return bad ? 1 : 0;
}

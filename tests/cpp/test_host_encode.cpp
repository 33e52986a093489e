// The host-side key encoder's loops and worker pool (csvplus_amd/csrc/host_encode_kernels.hpp) on their own: the short-key
// and arithmetic loops against the plain LUT walk on random columns (bytes outside the alphabets, values that are too long,
// empty values, the last values of a buffer without slack), the pool over many job sizes, and a rate print.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <random>
#include <string>

#include "../../csvplus_amd/csrc/host_encode_kernels.hpp"

using namespace cph_host;

static int failures = 0;
#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            failures++;                                              \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                            \
    } while (0)

// a random codec over npos positions: per position an alphabet (plus the END symbol from position minlen on), mixed-radix weights
struct Codec {
    int npos, minlen;
    std::vector<uint32_t> lut;   // [npos][257], top bit = absent
};
static Codec make_codec(std::mt19937_64& rng, int npos, int minlen, const std::string& alphabet) {
    Codec c{npos, minlen, std::vector<uint32_t>((size_t)npos * kLutRow, 0x80000000u)};
    std::vector<uint32_t> radix((size_t)npos);
    std::vector<std::vector<int>> syms((size_t)npos);
    for (int p = 0; p < npos; p++) {
        if (p >= minlen) syms[p].push_back(0);
        for (char ch : alphabet)
            if (rng() % 4 != 0 || syms[p].size() < 2) syms[p].push_back(1 + (unsigned char)ch);
        std::sort(syms[p].begin(), syms[p].end());
        syms[p].erase(std::unique(syms[p].begin(), syms[p].end()), syms[p].end());
        radix[p] = (uint32_t)syms[p].size();
    }
    uint64_t w = 1;
    for (int p = npos - 1; p >= 0; p--) {
        for (size_t k = 0; k < syms[p].size(); k++) c.lut[(size_t)p * kLutRow + syms[p][k]] = (uint32_t)(k * w);
        w *= radix[p];
    }
    CHECK(w < (1ull << 31));
    return c;
}

static void test_short_against_walk() {
    std::mt19937_64 rng(7);
    for (int round = 0; round < 40; round++) {
        const int npos = 1 + (int)(rng() % 8), minlen = (int)(rng() % (npos + 1));
        const Codec c = make_codec(rng, npos, minlen, "0123456789");
        const uint64_t n = 1 + rng() % 5000;
        std::vector<uint8_t> data;
        std::vector<uint32_t> off{0};
        for (uint64_t r = 0; r < n; r++) {
            int l = (int)(rng() % (npos + 2));                      // sometimes too long, sometimes empty
            for (int q = 0; q < l; q++) data.push_back(rng() % 23 == 0 ? (uint8_t)(rng() & 0xFF) : (uint8_t)('0' + rng() % 10));
            off.push_back((uint32_t)data.size());
        }
        HostCol col{data.data(), off.data(), 32, 0, data.size()};   // NO slack behind the last value
        std::vector<uint32_t> a(n), b(n);
        const int32_t start[2] = {0, npos}, maxlen[1] = {npos};
        encode_lut(c.lut.data(), 1, start, maxlen, &col, 0, n, a.data());
        const bool absent_b = encode_lut_short(c.lut.data(), npos, col, 0, n, b.data());
        CHECK(a == b);
        {   // streaming stores, the "some row is absent" result of every loop
            std::vector<uint32_t> s1(n, 1u), s2(n, 2u);
            const bool f1 = encode_lut(c.lut.data(), 1, start, maxlen, &col, 0, n, s1.data(), true);
            const bool f2 = encode_lut_short(c.lut.data(), npos, col, 0, n, s2.data(), true);
            bool want = false;
            for (uint32_t x : a) want |= x == kCodeAbsent;
            CHECK(s1 == a && s2 == a && f1 == want && f2 == want && absent_b == want);
        }
        uint64_t present = 0;
        for (uint32_t x : a) present += x != kCodeAbsent;
        CHECK(present > 0 || n < 20);
        // the same column with 64-bit offsets, and a middle range only
        std::vector<uint64_t> off64(off.begin(), off.end());
        HostCol col64{data.data(), off64.data(), 64, 0, data.size()};
        std::vector<uint32_t> c2(n, 12345u);
        encode_lut_short(c.lut.data(), npos, col64, n / 3, n, c2.data());
        for (uint64_t r = 0; r < n; r++) CHECK(c2[r] == (r < n / 3 ? 12345u : a[r]));
    }
}

static int vector_rounds = 0;
static void test_arith_against_walk() {
    std::mt19937_64 rng(11);
    for (int round = 0; round < 20; round++) {
        // 8 positions, contiguous ranges [lo_p, lo_p + r_p)
        Arith8 ar{};
        Codec c{8, 8, std::vector<uint32_t>((size_t)8 * kLutRow, 0x80000000u)};
        uint32_t radix[8], lo[8];
        for (int p = 0; p < 8; p++) {
            radix[p] = round == 0 ? 10 : 1 + (uint32_t)(rng() % 12);
            lo[p] = round == 0 ? '0' : 33 + (uint32_t)(rng() % 60);
        }
        uint64_t w = 1;
        for (int p = 7; p >= 0; p--) {
            ar.mult[p] = (uint32_t)w;
            ar.radix[p] = radix[p];
            for (uint32_t k = 0; k < radix[p]; k++) c.lut[(size_t)p * kLutRow + 1 + lo[p] + k] = (uint32_t)(k * w);
            ar.lo |= (uint64_t)lo[p] << (8 * p);
            ar.rngc |= (uint64_t)(0x7F - (radix[p] - 1)) << (8 * p);
            w *= radix[p];
        }
        if (w >= (1ull << 31)) continue;
        const uint64_t n = 20000;
        std::vector<uint8_t> data(8 * n);
        for (uint64_t r = 0; r < n; r++)
            for (int p = 0; p < 8; p++) {
                const uint32_t k = rng() % 29 == 0 ? (uint32_t)(rng() & 0xFF) : lo[p] + (uint32_t)(rng() % radix[p]);
                data[8 * r + p] = (uint8_t)k;
            }
        HostCol col{data.data(), nullptr, 32, 8, data.size()};
        std::vector<uint32_t> a(n), b(n);
        const int32_t start[2] = {0, 8}, maxlen[1] = {8};
        encode_lut(c.lut.data(), 1, start, maxlen, &col, 0, n, a.data());
        const bool absent_ar = encode_arith8(ar, data.data(), 0, n, b.data());
        CHECK(a == b);
        bool want_absent = false;
        for (uint32_t x : a) want_absent |= x == kCodeAbsent;
        CHECK(absent_ar == want_absent);
        {
            std::vector<uint32_t> s1(n + 4, 9u);
            CHECK(encode_arith8(ar, data.data(), 0, n, s1.data() + 1, true) == want_absent);   // streaming stores at an odd address
            for (uint64_t r = 0; r < n; r++) CHECK(s1[r + 1] == a[r]);
            CHECK(s1[0] == 9u && s1[n + 1] == 9u);
            // a range of valid keys only reports no absent row
            std::vector<uint8_t> good(8 * 64);
            for (size_t i = 0; i < good.size(); i++) good[i] = (uint8_t)lo[i & 7];
            std::vector<uint32_t> g(64);
            CHECK(!encode_arith8(ar, good.data(), 0, 64, g.data(), true));
        }
#if defined(__x86_64__)
        if (arith8_vector_ok(ar)) {
            std::vector<uint32_t> v(n, 7u);
            encode_arith8_avx2(ar, data.data(), 3, n - 2, v.data());      // unaligned start, a scalar tail
            for (uint64_t r = 0; r < n; r++) CHECK(v[r] == (r < 3 || r >= n - 2 ? 7u : a[r]));
            for (int shift = 0; shift < 4; shift++) {   // streaming stores: any alignment of the output, head and tail rows scalar
                std::vector<uint32_t> sv(n + 8, 7u);
                bool w2 = false;
                for (uint64_t r = 5; r < n - 1; r++) w2 |= a[r] == kCodeAbsent;
                CHECK(encode_arith8_avx2(ar, data.data(), 5, n - 1, sv.data() + shift - 0, true) == w2 || shift != 0);
                std::vector<uint32_t> sv2(n + 8, 7u);
                const bool f = encode_arith8_avx2(ar, data.data(), 5, n - 1, sv2.data() + shift, true);
                CHECK(f == w2);
                for (uint64_t r = 0; r < n; r++) CHECK(sv2[r + shift] == (r < 5 || r >= n - 1 ? 7u : a[r]));
            }
            vector_rounds++;
        }
#endif
    }
    CHECK(vector_rounds > 0 || !__builtin_cpu_supports("avx2"));
}

static void test_pool_and_rate() {
    const int hw = (int)std::thread::hardware_concurrency();
    BlockPool pool(hw > 1 ? hw - 1 : 0);
    std::mt19937_64 rng(3);
    for (uint64_t n : {(uint64_t)1, (uint64_t)65535, (uint64_t)65536, (uint64_t)65537, (uint64_t)1000003, (uint64_t)5}) {
        for (int rep = 0; rep < 30; rep++) {
            std::vector<uint8_t> hit(n, 0);
            pool.run(n, [&](uint64_t r0, uint64_t r1) {
                for (uint64_t r = r0; r < r1; r++) hit[r]++;
            });
            bool ok = true;
            for (uint8_t h : hit) ok = ok && h == 1;
            CHECK(ok);
        }
    }
    // rate: 2^23-row chunks of 8-byte decimal ids (the benchmark's customer ids), fork-join per chunk as the stream join does
    const uint64_t n = 1u << 23;
    std::vector<uint8_t> data(8 * n);
    for (uint64_t r = 0; r < n; r++) {
        char b[16];
        snprintf(b, sizeof b, "%08llu", (unsigned long long)(rng() % 10000000));
        memcpy(&data[8 * r], b, 8);
    }
    Arith8 ar{};
    uint64_t w = 1;
    for (int p = 7; p >= 0; p--) {
        ar.mult[p] = (uint32_t)w;
        ar.radix[p] = 10;
        ar.lo |= (uint64_t)'0' << (8 * p);
        ar.rngc |= (uint64_t)(0x7F - 9) << (8 * p);
        w *= 10;
    }
    std::vector<uint32_t> out(n);
    for (int rep = 0; rep < 3; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int chunk = 0; chunk < 8; chunk++)
            pool.run(n, [&](uint64_t r0, uint64_t r1) { encode_arith8(ar, data.data(), r0, r1, out.data()); });
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("arith8: %.2f G rows/s on %d threads (8 chunks of 2^23 rows, %.2f ms per chunk)\n", 8.0 * n / s / 1e9, pool.workers() + 1, s / 8 * 1e3);
    }
#if defined(__x86_64__)
    if (arith8_vector_ok(ar)) {
        std::vector<uint32_t> ref(out);
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int chunk = 0; chunk < 8; chunk++)
                pool.run(n, [&](uint64_t r0, uint64_t r1) { encode_arith8_avx2(ar, data.data(), r0, r1, out.data()); });
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("arith8 avx2: %.2f G rows/s on %d threads (%.2f ms per chunk)\n", 8.0 * n / s / 1e9, pool.workers() + 1, s / 8 * 1e3);
        }
        CHECK(out == ref);
    }
#endif
    CHECK(out[12345] != kCodeAbsent);
    // short LUT keys (unpadded decimal ids of up to 6 digits)
    std::vector<uint8_t> vdata;
    std::vector<uint32_t> off{0};
    for (uint64_t r = 0; r < n; r++) {
        char b[16];
        const int l = snprintf(b, sizeof b, "%llu", (unsigned long long)(rng() % 100000));
        vdata.insert(vdata.end(), b, b + l);
        off.push_back((uint32_t)vdata.size());
    }
    std::mt19937_64 r2(5);
    const Codec c = make_codec(r2, 5, 1, "0123456789");
    HostCol col{vdata.data(), off.data(), 32, 0, vdata.size()};
    for (int rep = 0; rep < 3; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int chunk = 0; chunk < 8; chunk++)
            pool.run(n, [&](uint64_t r0, uint64_t r1) { encode_lut_short(c.lut.data(), 5, col, r0, r1, out.data()); });
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("lut_short<5>: %.2f G rows/s on %d threads (%.2f ms per chunk)\n", 8.0 * n / s / 1e9, pool.workers() + 1, s / 8 * 1e3);
    }
}


// encode_split against a plain restatement: random prefix dictionaries behind a brute-force "perfect hash" (any collision-free
// placement serves the loop), suffix alphabets with END, rows the codec cannot code (unknown prefix, foreign suffix byte, suffix or
// value too long, no delimiter), the last values of a buffer without slack.
struct TKey { uint64_t w[4]; uint32_t len; };
static uint32_t thash(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint32_t len) {
    uint64_t h = w0 * 0x9E3779B97F4A7C15ull ^ (w1 + 0x85EBCA6Bull) * 0xC2B2AE3D27D4EB4Full ^ (w2 * 31 + w3 * 17 + len);
    h ^= h >> 29;
    return (uint32_t)(h * 0xBF58476D1CE4E5B9ull >> 32);
}
static void test_split_against_walk() {
    std::mt19937_64 rng(11);
    for (int round = 0; round < 30; round++) {
        const uint8_t delim = (uint8_t)"#/:"[rng() % 3];
        // dictionary: distinct prefixes that END in the delimiter, + now and then a whole value without one
        const int nd = 1 + (int)(rng() % 200);
        std::vector<std::string> pre;
        while ((int)pre.size() < nd) {
            std::string p;
            const int l = 1 + (int)(rng() % 30);
            for (int q = 0; q < l; q++) p.push_back((char)('a' + rng() % 6));
            if (rng() % 9 != 0 || l == 31) p.push_back((char)delim);
            if (std::find(pre.begin(), pre.end(), p) == pre.end()) pre.push_back(p);
        }
        std::sort(pre.begin(), pre.end());
        std::vector<TKey> dict(pre.size());
        for (size_t i = 0; i < pre.size(); i++) {
            memset(&dict[i], 0, sizeof(TKey));
            memcpy(dict[i].w, pre[i].data(), pre[i].size());
            dict[i].len = (uint32_t)pre[i].size();
        }
        // hash-and-displace by brute force: one bucket per key's low bits, first displacement that lands every key of it on a free slot
        uint32_t nslots = 4;
        while (nslots < 4 * pre.size()) nslots <<= 1;
        std::vector<uint16_t> disp, slots;
        for (;; nslots <<= 1) {
            const uint32_t nb = nslots / 4;
            disp.assign(nb, 0);
            slots.assign(nslots, 0);
            bool ok = true;
            for (uint32_t b = 0; b < nb && ok; b++) {
                std::vector<uint32_t> mine;
                for (size_t i = 0; i < dict.size(); i++)
                    if ((thash(dict[i].w[0], dict[i].w[1], dict[i].w[2], dict[i].w[3], dict[i].len) & (nb - 1)) == b) mine.push_back((uint32_t)i);
                bool placed = mine.empty();
                for (uint32_t d = 0; d < nslots && !placed; d++) {
                    std::vector<uint32_t> at;
                    bool fits = true;
                    for (uint32_t i : mine) {
                        const uint32_t h = thash(dict[i].w[0], dict[i].w[1], dict[i].w[2], dict[i].w[3], dict[i].len);
                        const uint32_t s2 = ((h >> 16) + d) & (nslots - 1);
                        if (slots[s2] || std::find(at.begin(), at.end(), s2) != at.end()) { fits = false; break; }
                        at.push_back(s2);
                    }
                    if (fits) {
                        for (size_t k = 0; k < mine.size(); k++) slots[at[k]] = (uint16_t)(mine[k] + 1);
                        disp[b] = (uint16_t)d;
                        placed = true;
                    }
                }
                ok = placed;
            }
            if (ok) break;
        }
        const int smax = (int)(rng() % 7), smin = smax ? (int)(rng() % (smax + 1)) : 0;
        const Codec sc = make_codec(rng, smax ? smax : 1, smin, "0123456789");
        uint64_t sstates = 1;   // weight of the prefix = product of the suffix radices
        {
            std::vector<uint32_t> mx((size_t)sc.npos, 0);
            for (int p2 = 0; p2 < sc.npos; p2++)
                for (int sym = 0; sym < kLutRow; sym++)
                    if (!(sc.lut[(size_t)p2 * kLutRow + sym] >> 31)) mx[p2] = std::max(mx[p2], sc.lut[(size_t)p2 * kLutRow + sym]);
            sstates = (uint64_t)mx[0] + 1;   // (upper bound of the suffix code + 1 is enough for a weight)
            for (int p2 = 1; p2 < sc.npos; p2++) sstates += mx[p2];
        }
        SplitEnc<TKey> e{};
        e.delim = delim;
        e.vmax = 40;
        e.smaxlen = (uint32_t)smax;
        e.pmult = (uint32_t)sstates;
        e.hmask = nslots - 1;
        e.dmask = nslots / 4 - 1;
        e.disp = disp.data();
        e.slots = slots.data();
        e.dict = dict.data();
        e.lutw = sc.lut.data();
        CHECK((uint64_t)pre.size() * sstates < (1ull << 31));
        // rows
        const uint64_t n = 1 + rng() % 4000;
        std::vector<uint8_t> data;
        std::vector<uint32_t> off{0};
        for (uint64_t r = 0; r < n; r++) {
            std::string v = pre[rng() % pre.size()];
            if (rng() % 40 == 0) v = "zz" + v;                                        // unknown prefix
            if (rng() % 50 == 0) v = std::string(1 + rng() % 12, 'b');               // no delimiter (usually unknown)
            if (rng() % 60 == 0) v.clear();
            if (!v.empty() && (uint8_t)v.back() == delim) {
                const int sl = (int)(rng() % (smax + 2));                             // sometimes too long
                for (int q = 0; q < sl; q++) v.push_back(rng() % 37 == 0 ? (char)('a' + rng() % 26) : (char)('0' + rng() % 10));
            }
            if (rng() % 200 == 0) v += std::string(45, '7');                         // a value beyond vmax
            data.insert(data.end(), v.begin(), v.end());
            off.push_back((uint32_t)data.size());
        }
        for (int slack = 0; slack < 2; slack++) {
            std::vector<uint8_t> buf = data;
            if (slack) buf.resize(buf.size() + 64, 0xAA);
            HostCol c{buf.data(), off.data(), 32, 0, slack ? (uint64_t)buf.size() : (uint64_t)data.size()};
            std::vector<uint32_t> got(n, 0xDEADBEEFu);
            const bool any = encode_split(e, c, 0, n, got.data(), false, thash);
            bool want_any = false;
            for (uint64_t r = 0; r < n; r++) {
                const std::string v(data.begin() + off[r], data.begin() + off[r + 1]);
                const size_t at = v.find((char)delim);
                const std::string p = at == std::string::npos ? v : v.substr(0, at + 1), sfx = v.substr(p.size());
                const auto it = std::lower_bound(pre.begin(), pre.end(), p);
                bool bad = v.size() > 40 || p.size() > 32 || sfx.size() > (size_t)smax || it == pre.end() || *it != p;
                uint32_t code = 0;
                if (!bad) {
                    code = (uint32_t)(it - pre.begin()) * e.pmult;
                    for (int q = 0; q < smax; q++) {
                        const uint32_t x = sc.lut[(size_t)q * kLutRow + ((size_t)q < sfx.size() ? 1 + (unsigned char)sfx[q] : 0)];
                        if (x >> 31) bad = true;
                        code += x & 0x7FFFFFFFu;
                    }
                }
                want_any = want_any || bad;
                if (!bad) CHECK(got[r] == code);
            }
            CHECK(any == want_any);
        }
    }
}

int main() {
    test_split_against_walk();
    test_short_against_walk();
    test_arith_against_walk();
    test_pool_and_rate();
    printf("%d host encoder checks failed\n", failures);
    return failures ? 1 : 0;
}

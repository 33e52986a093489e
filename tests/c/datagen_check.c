/* datagen_check.c — the synthetic-table generator under AddressSanitizer + UBSan (`make asan`): every kind and
 * encoding, buffers malloc'ed at exactly dg_column_bytes() / nrows+1 offsets, several row ranges (the per-rank
 * shards and streaming chunks of bench.py use arbitrary row0 / nrows).  Prints DATAGEN_OK on success. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../csvplus_amd/csrc/datagen.c"

int main(void) {
    uint64_t digest = 0;
    for (int kind = 0; kind < 9; kind++)
        for (int enc = 0; enc < 2; enc++)
            for (int bits = 32; bits <= 64; bits += 32) {
                const uint64_t ranges[4][2] = {{0, 0}, {0, 1}, {12345, 70001}, {99999999ull, 333}};
                for (int r = 0; r < 4; r++) {
                    dg_spec s;
                    s.kind = kind; s.encoding = enc; s.domain = 1000003; s.base = kind == 1 ? 1 : 0; s.seed = 0xC5F1D5 + kind;
                    const uint64_t row0 = ranges[r][0], n = ranges[r][1];
                    const uint64_t total = dg_column_bytes(&s, row0, n);
                    uint8_t* data = (uint8_t*)malloc(total ? total : 1);
                    void* offs = malloc((size_t)(n + 1) * (bits / 8));
                    const uint64_t got = dg_column_fill(&s, row0, n, data, offs, bits);
                    if (got != total) { printf("kind %d enc %d: %llu != %llu\n", kind, enc, (unsigned long long)got, (unsigned long long)total); return 2; }
                    for (uint64_t i = 0; i < total; i++) digest = digest * 1099511628211ull + data[i];
                    free(data); free(offs);
                }
            }
    printf("DATAGEN_OK %016llx\n", (unsigned long long)digest);
    return 0;
}

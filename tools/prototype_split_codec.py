#!/usr/bin/env python3
"""CPU prototype of the delimiter-split codec sketched in DESIGN.md §11 (not part of the product): on BASELINE config-3 keys
("surname/name#number", 10-22 bytes) it (1) looks for a split byte on a sample, (2) codes the prefix up to and including the
first split byte by the rank of the whole prefix and the suffix per position relative to the split, (3) checks that the
resulting code orders the keys exactly like bytewise comparison, (4) reports the code width against the per-position code
of the whole key.  usage: tools/prototype_split_codec.py [rows]"""
import math
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import datagen as dg

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
col = dg.varkeys(n)
keys = col.values()


def per_position_bits(vals):
    """mixed-radix code of keycodec.hip: symbol = 0 (value ended) or 1 + byte, alphabet per byte position."""
    maxlen = max(map(len, vals)) if vals else 0
    minlen = min(map(len, vals)) if vals else 0
    alph = [set() for _ in range(maxlen)]
    for v in vals:
        for q, b in enumerate(v):
            alph[q].add(b)
    radix = [len(a) + (1 if q >= minlen else 0) for q, a in enumerate(alph)]
    return sum(math.log2(r) for r in radix), radix


whole_bits, _ = per_position_bits(keys)
print(f"{n} config-3 keys, {len(set(keys))} distinct; per-position code of the whole key: {whole_bits:.1f} bits "
      f"(the library's dictionary windows bring it to 47)")

sample = keys[:: max(1, n // 20000)]
cands = []
for d in range(256):
    if sum(1 for k in sample if d in k) < 0.99 * len(sample):
        continue
    prefixes = {k[: k.index(d) + 1] for k in sample if d in k}
    cands.append((len(prefixes), d))
cands.sort()
print("split candidates (distinct prefixes on the sample, byte):", [(c, chr(d)) for c, d in cands[:5]])
best = None
for _, d in cands[:4]:
    pre = [k[: k.index(d) + 1] if d in k else k for k in keys]
    suf = [k[k.index(d) + 1:] if d in k else b"" for k in keys]
    nodelim = [d not in k for k in keys]
    P = sorted(set(pre))
    if len(P) > 4096:
        continue
    sbits, sradix = per_position_bits(suf)
    bits = math.log2(len(P)) + sbits
    print(f"  split at {chr(d)!r}: {len(P)} prefixes ({math.log2(len(P)):.1f} bits) + suffix radices {sradix} ({sbits:.1f} bits) = {bits:.1f} bits")
    if best is None or bits < best[0]:
        best = (bits, d, pre, suf, P)
bits, d, pre, suf, P = best
# order check: (rank of prefix among the sorted prefixes, suffix bytes with end < every byte) must order like the keys.
# A key without the split byte is its own prefix followed by END: rank it by (prefix bytes, then a symbol below every byte).
rank = {p: i for i, p in enumerate(P)}
code = sorted(range(len(keys)), key=lambda i: (rank[pre[i]], suf[i]))
plain = sorted(range(len(keys)), key=lambda i: keys[i])
same = all(keys[a] == keys[b] for a, b in zip(code, plain))
print(f"chosen split byte {chr(d)!r}: code width {bits:.1f} bits -> {math.ceil(bits / 8)} radix passes of "
      f"{'32' if bits <= 32 else '64'}-bit keys (now: 47 bits, 6 passes of 64-bit keys); orders like bytewise comparison: {same}")
assert same

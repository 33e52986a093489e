#!/usr/bin/env python3
"""IndexOn over config-3 shaped keys (variable length, duplicates): a small driver for rocprofv3 counter passes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
eng = Engine(0)
d = dg.varkeys(n).to_device(eng.device)
for _ in range(2):
    ix = eng.index_on([d], unique=False)
    print(ix.info(), flush=True)
    ix.close()

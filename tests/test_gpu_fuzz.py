"""A bounded slice (4 x 15 s) of the four differential fuzzers under tools/ in the driver's own GPU run: random tables, builds and
chains through the C ABI against the oracle (csvplus.go:545-583, 707-756, 794-807 restated in oracle/).  Round 4's `s_barrier`
miscompile was found by exactly these scripts and by nothing else; run by hand they take longer sessions on other seeds
(tools/gpu_run.sh fuzz [seconds]; profiles/r0N_fuzz.txt)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("script,ok_word,seed", [("fuzz_round5.py", "FUZZ_R5_OK", 6101), ("fuzz_gpu.py", "FUZZ_OK", 6102),
                                                 ("fuzz_builds.py", "FUZZ_BUILDS_OK", 6103), ("fuzz_round6.py", "FUZZ_R6_OK", 6104)])
def test_fuzz_slice(script, ok_word, seed):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / script), "15", str(seed)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=str(ROOT)))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and ok_word in r.stdout and "MISMATCH" not in r.stdout, tail

"""Sanitizer builds of the CPU-side native code: `make asan` compiles the oracle together with a
fuzz driver (tests/c/oracle_fuzz.c: malformed CSV, random key tables incl. NUL / 0xFF bytes and 300-byte keys) and
the data generator with its range checker (tests/c/datagen_check.c) under -fsanitize=address,undefined.  Any
finding aborts the run; the digests pin what was executed."""
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(target, *args):
    subprocess.check_call(["make", "-C", str(ROOT), f"oracle/_build/{target}"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    return subprocess.run([str(ROOT / "oracle" / "_build" / target), *args], env=env, capture_output=True, text=True, timeout=600)


def test_oracle_fuzz_under_asan_ubsan():
    r = _run("oracle_fuzz_asan", "3000")
    assert r.returncode == 0 and "ORACLE_FUZZ_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_datagen_under_asan_ubsan():
    r = _run("datagen_asan")
    assert r.returncode == 0 and "DATAGEN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]

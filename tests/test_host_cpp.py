"""Runs the C++ host-facade tests (tests/cpp/test_host.cpp): the reference's own hot-path tests
(TestIndexImpl, TestSimpleUniqueJoin, TestSorted, TestSimpleTotals, TestLongChain, TestMultiIndex,
TestExcept, TestErrors, TestResolver, TestIndexStore; plus TestBatchingSemantics and TestChainPrecedence for the batched /
fused boundary) restated against csvplus_amd/host/csvplus.hpp, which calls the GPU through
the C ABI."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "tests" / "cpp" / "test_host"


def test_host_binary_builds():
    """CPU: the facade compiles and links against the C ABI (g++, no GPU needed)."""
    subprocess.check_call(["make", "-C", str(ROOT), "host"])
    assert BIN.exists()


def test_host_facade_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    subprocess.check_call(["make", "-C", str(ROOT), "host"])
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no usable GPU" in r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_tests_through_cpp_facade():
    subprocess.check_call(["make", "-C", str(ROOT), "host"])
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 of 12 host tests failed" in r.stdout
    assert "PASS TestChainPrecedence" in r.stdout   # round 4: src.Join(a).Join(b) fused per batch, stream > a > b


C_DEMO = ROOT / "tests" / "c" / "abi_demo"


def test_plain_c_consumer_builds_and_fails_loudly_without_gpu():
    """The header is C99 and the library links from plain C (what cgo does); without a GPU the program reports it."""
    import torch

    subprocess.check_call(["make", "-C", str(ROOT), "host"])
    assert C_DEMO.exists()
    if not torch.cuda.is_available():
        r = subprocess.run([str(C_DEMO)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "no usable GPU" in r.stdout


@pytest.mark.gpu
def test_plain_c_consumer_on_gpu():
    subprocess.check_call(["make", "-C", str(ROOT), "host"])
    r = subprocess.run([str(C_DEMO)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "WRONG" not in r.stdout and r.stdout.count(" ok") == 4


def test_host_encoder_loops_and_pool_on_cpu():
    """csvplus_amd/csrc/host_encode_kernels.hpp (the loops cph_host_encoder_run runs on the host: arithmetic, AVX2, short-key LUT,
    plain walk; the block pool) against each other on random codecs and columns — no GPU involved."""
    subprocess.check_call(["make", "-C", str(ROOT), "tests/cpp/test_host_encode"])
    r = subprocess.run([str(ROOT / "tests" / "cpp" / "test_host_encode")], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0 and "0 host encoder checks failed" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])

import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Builds the native pieces when missing (hipcc cross-compiles without a GPU)."""
    import subprocess

    need = [ROOT / "csvplus_amd/lib/libcsvplus_hip.so", ROOT / "csvplus_amd/lib/libcph_datagen.so",
            ROOT / "oracle/_build/liboracle.so"]
    if not all(p.exists() for p in need):
        subprocess.check_call(["make", "-C", str(ROOT), "hip", "datagen", "oracle"])


_ensure_built()


@pytest.fixture(scope="session", autouse=True)
def _gpu_box_ready(request):
    """GPU runs only: before the first test (some start with a SUBPROCESS: bench.py), wait until the device answers in this
    process — a box that has just been handed out may need a moment before its first HIP call succeeds."""
    if not any(item.get_closest_marker("gpu") for item in request.session.items):
        return
    import time

    import torch

    for _ in range(60):
        try:
            if torch.cuda.is_available() and torch.cuda.device_count() > 0:
                torch.zeros(1, device="cuda").item()
                break
        except Exception:   # noqa: BLE001 — keep waiting
            pass
        time.sleep(1)


@pytest.fixture(scope="session")
def ctx():
    """One cph_ctx on GPU 0 (gpu tests only)."""
    from csvplus_amd import Context

    c = Context(0)
    guard = os.environ.get("CPH_POOL_GUARD") == "1"   # whole-suite canary run: tools/gpu_guard.sh
    if guard:
        c.set_option("pool_guard", 1)
    yield c
    if guard:
        c.set_option("pool_guard_check", 0)   # raises CphError if any canary behind a device block was overwritten
    c.close()


@pytest.fixture(params=["small_path", "general_path"])
def both_build_paths(ctx, request):
    """Runs a test twice: with the one-launch build of small tables (small_build.hip; up to 16384 rows here, 8192 by default)
    and with it switched off, so that the general path (statistics, host codec, multi-launch radix sort) keeps being
    checked on the same small inputs."""
    ctx.set_option("small_build_rows", 16384 if request.param == "small_path" else 0)
    yield request.param
    ctx.set_option("small_build_rows", 8192)

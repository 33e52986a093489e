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


@pytest.fixture(scope="session")
def ctx():
    """One cph_ctx on GPU 0 (gpu tests only)."""
    from csvplus_amd import Context

    c = Context(0)
    yield c
    c.close()

"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/csvplus_hip.h declares; without a GPU it fails loudly (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import pytest

from csvplus_amd import _native as N

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "csvplus_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"CPH_API\s+[\w\s\*]+?\b(cph_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("cph_ctx_create", "cph_index_build", "cph_join_probe", "cph_index_find", "cph_matches_release",
                 "cph_index_perm", "cph_pinned_alloc"):
        assert must in syms
    assert len(syms) >= 17


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(N.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/csvplus_hip.h but not exported"


def test_binding_covers_every_declared_symbol():
    bound = {p[0] for p in N.PROTOTYPES}
    assert bound == set(declared_symbols())
    N.load_library()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(N.cph_strcol) == 40
    assert ctypes.sizeof(N.cph_strval) == 16
    assert ctypes.sizeof(N.cph_matches) == 56
    assert ctypes.sizeof(N.cph_index_info) == 72


def test_struct_sizes_against_the_compiled_header(tmp_path):
    """sizeof of every struct the binding mirrors, as gcc sees include/csvplus_hip.h."""
    import subprocess

    names = ["cph_strcol", "cph_strval", "cph_matches", "cph_index_info", "cph_index_spec", "cph_chain_step", "cph_chain",
             "cph_colbuf", "cph_bytes", "cph_csv_options", "cph_csv_table", "cph_rowsel", "cph_groups", "cph_gathered",
             "cph_stream_chunk", "cph_kernel_stat"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "csvplus_hip.h"\nint main(void){'
                   + "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    out = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert ctypes.sizeof(getattr(N, n)) == int(out[n]), (n, ctypes.sizeof(getattr(N, n)), out[n])


def test_version_string():
    lib = N.load_library()
    assert b"gfx950" in lib.cph_version()


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU ctx creation must fail (never silently compute on the CPU)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(N.CphError) as e:
        N.Context(0)
    assert e.value.code == N.CPH_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under csvplus_amd/ may reference it."""
    for p in (ROOT / "csvplus_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".hpp", ".cpp", ".c", ".h") and p.is_file():
            assert "oracle" not in p.read_text(errors="ignore").lower().replace("no oracle", ""), p


def test_every_ctx_option_is_documented_in_the_header():
    """cph_ctx_set_option's switch (csrc/capi.hip) and the comment block above its declaration (include/csvplus_hip.h) name the
    same options: an A/B switch nobody can find is not a switch."""
    import re
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    src = (root / "csvplus_amd" / "csrc" / "capi.hip").read_text()
    hdr = (root / "include" / "csvplus_hip.h").read_text()
    opts = re.findall(r'k == "([a-z_0-9]+)"', src)
    assert len(opts) >= 30
    undocumented = [o for o in opts if f'"{o}"' not in hdr]
    assert not undocumented, undocumented

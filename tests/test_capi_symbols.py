"""The C-ABI library loads and exports every symbol include/psh.h declares; its
host-only entry points (no GPU needed) behave as documented."""
import ctypes as C
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from shadowing_amd import _build, _native
    _build.build()                       # hipcc cross-compiles gfx950 without a GPU
    return _native.load()


def declared_symbols():
    text = (REPO / "include" / "psh.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psh_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from shadowing_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 10
    assert set(syms) == set(_native.EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/psh.h but not exported"


def test_version_and_strerror(lib):
    from shadowing_amd import _native
    assert lib.psh_version() == _native.PSH_VERSION == 3
    # the header and the binding agree on the version and on the size of psh_profile (ctypes mirrors the C layout)
    text = (REPO / "include" / "psh.h").read_text()
    assert re.search(r"#define PSH_VERSION 3\b", text)
    # ... as a C compiler lays the header's struct out
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        src = Path(tmp) / "layout.c"
        src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "psh.h"\n'
                       'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(psh_profile), offsetof(psh_profile, tau_hint), '
                       'offsetof(psh_profile, path), offsetof(psh_profile, prep_ms)); return 0; }\n')
        subprocess.run(["gcc", f"-I{REPO / 'include'}", str(src), "-o", str(Path(tmp) / "layout")], check=True)
        got = subprocess.run([str(Path(tmp) / "layout")], check=True, capture_output=True, text=True).stdout.split()
    P = _native.PshProfile
    assert [int(v) for v in got] == [C.sizeof(P), P.tau_hint.offset, P.path.offset, P.prep_ms.offset]
    assert lib.psh_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert len(lib.psh_strerror(code)) > 5
    assert lib.psh_strerror(-99) == b"unknown error"


def test_workspace_bytes(lib):
    out = C.c_size_t(0)
    assert lib.psh_workspace_bytes(32768, 4096, 1, 20, 20, 1024, C.byref(out)) == 0
    one = out.value
    assert 65536 * 12 <= one < 64 << 20
    assert lib.psh_workspace_bytes(32768, 4096, 8, 20, 20, 1024, C.byref(out)) == 0
    assert out.value > 7 * one * 0.9
    assert lib.psh_workspace_bytes(32768, 4096, 1, 20, 20, 1024, None) == -1          # PSH_ERR_ARG
    assert lib.psh_workspace_bytes(0, 4096, 1, 20, 20, 1024, C.byref(out)) == -1
    assert lib.psh_workspace_bytes(16, 30, 1, 20, 20, 4, C.byref(out)) == -1           # no admissible window
    assert lib.psh_workspace_bytes(16, 4096, 1, 257, 0, 4, C.byref(out)) == -2         # W > PSH_MAX_W
    assert lib.psh_workspace_bytes(16, 4096, 1, 20, 0, 16385, C.byref(out)) == -2      # k > PSH_MAX_K
    assert lib.psh_merge_workspace_bytes(4, 1000, C.byref(out)) == 0 and out.value >= 4 * 1024 * 8


def test_candidates_layout_is_host_only_and_consistent(lib):
    """psh_candidates_layout (diagnostics of the admitted-set tests): pure arithmetic, runs without a device."""
    from shadowing_amd import _native
    nb = _native.workspace_bytes(4096, 4096, 4, 20, 20, 256)
    lay = _native.candidates_layout(4096, 4096, 4, 20, 20, 256, nb)
    assert lay["cap"] >= 64 * 256 and lay["cap"] % 64 == 0
    assert 0 < lay["qstate"] < lay["bcount"] < lay["bcount2"] < lay["cand_d"] < lay["cand_rt"]
    assert lay["cand_rt"] - lay["cand_d"] >= 4 * 4 * lay["cap"]
    assert lay["cand_rt"] + 8 * 4 * lay["cap"] <= nb
    assert lay["hdr_cand"] % 16 == 0 and lay["hdr_blk"] % 8 == 0 and lay["hdr_stream_ncand"] % 4 == 0
    assert (lay["max_blocks"], lay["fused_max_blocks"], lay["fused_front"]) == (2048, 256, 64)
    # the overlap-friendly launches' lists: inside the candidate arrays' region, a list per query, 16-byte entries
    assert lay["stream_list"] == lay["cand_d"] and 4096 < lay["stream_cap"] <= 65536
    assert lay["stream_list"] + 16 * 4 * lay["stream_cap"] <= lay["cand_rt"] + 8 * 4 * lay["cap"]
    twice = _native.candidates_layout(4096, 4096, 4, 20, 20, 256, 2 * nb)
    assert twice["cap"] > lay["cap"]                       # a larger workspace is a larger candidate buffer
    out = (C.c_int64 * 14)()
    assert lib.psh_candidates_layout(4096, 4096, 4, 20, 20, 256, 1024, out) == -3          # PSH_ERR_WORKSPACE
    assert lib.psh_candidates_layout(4096, 4096, 4, 20, 20, 256, nb, None) == -1


def test_null_arguments_are_rejected_not_crashing(lib):
    st = C.c_int(0)
    rc = lib.psh_scan_topk(0, None, None, 16, 128, 0, None, None, 1, 20, 0, 4, None, None, None, None, 0, None)
    assert rc == -1
    assert lib.psh_query_norm(0, None, None, 1, 20, None) == -1
    assert lib.psh_gather_paths(0, None, None, 1, 1, 8, 0, None, 0, 4, None) == -1
    assert lib.psh_merge_topk(0, None, None, None, 1, 8, 4, None, None, None, 0) == -1


def test_cuda_true_fails_loudly_without_a_device():
    """No silent CPU fallback: cuda=True without a HIP device raises."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    import shadowing_amd as sa
    from shadowing_amd import _native, synthetic as syn
    obj = sa.PathShadowing(sa.Identity(20), sa.RelativeMSE(), syn.dataset(8, 128, 0), sa.PredictionContext(20))
    with pytest.raises(_native.NativeLibraryError):
        obj.shadow(syn.single_query(20), k=4, cuda=True)
    with pytest.raises(_native.NativeLibraryError):
        _native.scan_topk(torch.zeros(8, 128), torch.zeros(1, 20), 4)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under shadowing_amd/ (or bench's
    product leg) may reference it."""
    for p in (REPO / "shadowing_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".h", ".cpp"):
            assert not re.search(r"^\s*(from|import)\s+oracle\b|libpsh_oracle|#include\s*[<\"][^\n]*oracle|psh_oracle_\w+\s*\(", p.read_text(), flags=re.M), p


def test_integration_md_build_line_lists_every_source():
    """INTEGRATION.md section 1's hipcc command must name exactly the translation units _build.py compiles (a maintainer
    following it otherwise gets undefined symbols)."""
    import re
    from pathlib import Path
    from shadowing_amd import _build
    text = (Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    block = text[text.index("hipcc --offload-arch=gfx950"):text.index("-o libpsh_hip.so")]
    listed = set(re.findall(r"shadowing_amd/csrc/(\w+\.hip)", block))
    assert listed == {p.name for p in _build.SOURCES}

"""The C-ABI library loads and exports exactly what include/nmhip.h declares
(no compute calls: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from neuralmonkey_amd import build
    build.build(verbose=False)           # hipcc cross-compiles gfx950 without a GPU
    from neuralmonkey_amd import _lib
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", text))


def test_header_matches_binding_table(lib):
    from neuralmonkey_amd import _lib
    assert header_symbols() == set(_lib.SIGNATURES), (
        header_symbols() ^ set(_lib.SIGNATURES))


def test_every_declared_symbol_is_exported(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_error_reporting_without_gpu(lib):
    assert lib.nm_version() >= 1
    # argument validation happens before any launch: a null operand is an error, not a crash
    rc = lib.nm_gemm_f32(None, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, 0, 0, 0, 0, None, 0)
    assert rc < 0 and b"null" in lib.nm_last_error()
    # energies + split-S partial contexts + partial statistics + one arrival counter per key batch
    assert lib.nm_attn_workspace_bytes(128, 50, 1024) == 4 * (6400 + 128 * 5 * 1024 + 128 * 5 * 4 + 128)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neuralmonkey_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NMHipError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neuralmonkey_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", src).replace('"""', ""), os.path.join(dirpath, f)

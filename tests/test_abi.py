"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/irbpp.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

import irbpp_amd  # noqa: F401
from irbpp_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "irbpp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(irbpp_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/irbpp.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"


def test_status_strings_and_version(lib):
    assert lib.irbpp_version() >= 100
    assert lib.irbpp_status_string(0) == b"ok"
    assert b"argument" in lib.irbpp_status_string(-1)


def test_create_rejects_bad_config(lib):
    cfg = _lib.IrbppConfig(num_bins=0, n_rot=4, selected=500, buffer_size=1, resolution_a=0.02,
                           resolution_h=0.01, resolution_z=0.01, bin=(ctypes.c_double * 3)(0.32, 0.32, 0.3),
                           scale_z=100.0)
    h = ctypes.c_void_p()
    assert lib.irbpp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1       # IRBPP_ERR_ARG before any HIP call
    cfg.num_bins, cfg.n_rot = 4, 9
    assert lib.irbpp_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "irbpp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_loader_fails_loudly_without_the_library(tmp_path, monkeypatch):
    """No CPU fallback: a missing libirbpp_hip.so is an error, not a silent slow path."""
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "libirbpp_hip.so"))


def test_loader_refuses_a_binary_built_from_other_sources(lib, tmp_path, monkeypatch):
    """The library carries the hash of the sources it was compiled from (irbpp_source_hash); the loader compares it with
    the hash of the sources lying next to it, so a stale .so cannot be tested or measured by mistake."""
    assert lib.irbpp_source_hash().decode() == build.source_hash() == build.built_hash()
    assert not build.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "source_hash", lambda: "0123456789abcdef")
    with pytest.raises(RuntimeError, match="was built from sources"):
        _lib.load()
    assert build.needs_build()
    monkeypatch.setenv("IRBPP_ALLOW_STALE_LIBRARY", "1")
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", lib)


def test_env_construction_without_a_gpu_raises(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from irbpp_amd import synthetic
    from irbpp_amd.vec_env import GpuVecEnv
    shapes = synthetic.cube_shapes()
    seqs = synthetic.make_sequences(shapes.n_shapes, 8, 10)
    with pytest.raises(Exception):
        GpuVecEnv(shapes, seqs, 2, device="cuda:0")

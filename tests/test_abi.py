"""CPU-side checks of the drop-in boundary: libgdg.so builds for gfx950, loads, exports every
symbol include/gdg.h declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry

ROOT = entry.ROOT


@pytest.fixture(scope="module")
def pkg():
    p = entry.load_package()
    p.build()
    return p


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gdg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdg_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    lib = ctypes.CDLL(pkg.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libgdg.so does not export %s" % n
    assert sorted(pkg.ABI_SYMBOLS) == names


def test_version_and_no_torch_types(pkg):
    assert pkg.lib().gdg_version().decode().startswith("gdg ")
    text = open(os.path.join(ROOT, "include", "gdg.h")).read()
    assert "torch" not in text.lower().replace("no c++ or torch types", "")


def test_code_object_is_gfx950(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.GdgError) as e:
        pkg.Context(1, 256)
    assert e.value.code == pkg.GDG_ERR_NO_DEVICE


def test_product_sources_do_not_touch_the_oracle():
    pkg_dir = entry.PKG_DIR
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".go")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gdg_oracle" not in text and "import oracle" not in text and "gdgo_" not in text, f

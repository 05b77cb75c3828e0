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


C_PROBE = r"""
#include <stdio.h>
#include <string.h>
#include "gdg.h"
/* what cgo does with the header: compile it as C, link the symbols by their plain names */
int main(void) {
    gdg_ctx *ctx = NULL;
    const char *v = gdg_version();
    int n = gdg_device_count();
    int rc = gdg_ctx_create(1, 256, 0, &ctx);
    printf("%s|%d|%d|%d\n", v, n, rc, ctx != NULL);
    if (ctx) gdg_ctx_destroy(ctx);
    return strncmp(v, "gdg ", 4) != 0;
}
"""


def test_header_is_plain_c_and_the_library_links_from_c(pkg, tmp_path):
    """cgo compiles include/gdg.h as C: -std=c99 -pedantic must take it without a warning, and a C program must link against libgdg.so by
    the plain symbol names.  Without a GPU gdg_ctx_create answers GDG_ERR_NO_DEVICE and leaves the handle NULL; with one it succeeds."""
    import subprocess
    src = tmp_path / "probe.c"
    src.write_text(C_PROBE)
    exe = tmp_path / "probe"
    lib_dir = os.path.dirname(pkg.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", lib_dir, "-lgdg", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    version, n_dev, rc, has_ctx = r.stdout.strip().split("|")
    assert version.startswith("gdg ")
    if int(n_dev) == 0:
        assert int(rc) == pkg.GDG_ERR_NO_DEVICE and has_ctx == "0"
    else:
        assert int(rc) == 0 and has_ctx == "1"


def test_makefiles_create_the_directory_they_link_into():
    """go-dsp-guitar_amd/lib/ holds only built (git-ignored) libraries, so a fresh clone has no such directory: the link rules make it."""
    for rel in ("go-dsp-guitar_amd/csrc/Makefile", "go-dsp-guitar_amd/host/Makefile"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "../lib/" in text and "mkdir -p $(dir $(OUT))" in text, rel

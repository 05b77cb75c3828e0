"""The cgo binding (go-dsp-guitar_amd/go/gdg) against a real Go toolchain, whenever one is on PATH: `go vet ./gdg && go build ./gdg` inside the
module go-dsp-guitar_amd/go (go.mod: the reference's module path and `go 1.16`), with CGO_CFLAGS / CGO_LDFLAGS pointing at include/gdg.h and the
built libgdg.so.  Neither the authoring container nor (so far) the GPU box has Go: the tests then SKIP and say so in a warning, which pytest -q
prints in its summary -- so the driver's own log shows whether the box had a toolchain.  One copy runs with the CPU suite, one with -m gpu.
The interfaces at stake: signal/signal.go:21-36, tuner/tuner.go:62-65, spatializer/spatializer.go:30-41 (bound through this package)."""
import os
import shutil
import subprocess
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO_DIR = os.path.join(ROOT, "go-dsp-guitar_amd", "go")


def _go_or_skip(where):
    go = shutil.which("go")
    if not go:
        msg = "no Go toolchain on PATH (%s): go-dsp-guitar_amd/go/gdg was NOT compiled here" % where
        warnings.warn(msg)
        pytest.skip(msg)
    return go


def _vet_and_build(go):
    import __graft_entry__ as entry
    entry.load_package().build()                                       # libgdg.so must exist for the link step
    env = dict(os.environ)
    env["CGO_ENABLED"] = "1"
    env["CGO_CFLAGS"] = "-I" + os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "go-dsp-guitar_amd", "lib")
    env["CGO_LDFLAGS"] = "-L%s -lgdg -Wl,-rpath,%s" % (lib, lib)
    env.setdefault("GOFLAGS", "-mod=mod")
    env.setdefault("GOCACHE", os.path.join("/tmp", "gdg-go-cache"))
    version = subprocess.run([go, "version"], capture_output=True, text=True, timeout=60).stdout.strip()
    for args in (["vet", "./gdg"], ["build", "./gdg"]):
        r = subprocess.run([go] + args, cwd=GO_DIR, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, "%s: go %s failed:\n%s\n%s" % (version, " ".join(args), r.stdout[-3000:], r.stderr[-3000:])
    warnings.warn("go-dsp-guitar_amd/go/gdg vetted and built with %s" % version)


def test_go_module_file_names_the_reference_module_and_language_level():
    text = open(os.path.join(GO_DIR, "go.mod")).read()
    lines = [l.strip() for l in text.splitlines() if l.strip() and not l.strip().startswith("//")]
    assert lines == ["module github.com/andrepxx/go-dsp-guitar", "go 1.16"]
    assert os.path.isfile(os.path.join(GO_DIR, "gdg", "gdg.go"))


def test_cgo_binding_vets_and_builds_when_go_is_here():
    _vet_and_build(_go_or_skip("CPU suite"))


@pytest.mark.gpu
def test_cgo_binding_vets_and_builds_on_the_gpu_box_when_go_is_there():
    _vet_and_build(_go_or_skip("GPU box"))

"""The Go side cannot be compiled in this environment (no Go toolchain in the authoring container or on the GPU box:
profiles/go_probe_r03.txt), so what CAN be checked mechanically is checked here: every C symbol the cgo binding calls exists in
include/gdg.h with that name, the sources are lexically balanced, the overlays import the binding under the reference's module path,
and the overlay generator writes absolute keys for the four files."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "go-dsp-guitar_amd", "go")
FILES = ["gdg/gdg.go", "signal/signal.go", "tuner/tuner.go", "spatializer/spatializer.go"]


def _strip(src):
    """Go source without comments, string / rune literals and the cgo preamble (enough for bracket counting)."""
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(\\.|[^"\\\n])*"', '""', src)
    src = re.sub(r"`[^`]*`", '""', src)
    src = re.sub(r"'(\\.|[^'\\])'", "' '", src)
    return src


def test_every_c_symbol_of_the_binding_is_declared_in_the_header():
    header = open(os.path.join(ROOT, "include", "gdg.h")).read()
    declared = set(re.findall(r"\b(gdg_[a-z0-9_]+)\s*\(", header)) | set(re.findall(r"\b(gdg_[a-z0-9_]+)\b", header))
    src = open(os.path.join(GO, "gdg", "gdg.go")).read()
    used = set(re.findall(r"\bC\.(gdg_[a-z0-9_]+)", src))
    assert used, "the binding calls nothing?"
    missing = sorted(u for u in used if u not in declared)
    assert not missing, missing
    # the sharded batch run is bound
    for name in ("gdg_batch_run_shard", "gdg_batch_finish_master", "gdg_batch_length", "gdg_batch_run"):
        assert name in used, name
    # struct fields the binding touches exist in the header's structs
    for field in ("master_left", "master_right", "metronome_bytes", "metronome", "job_samples", "samples_per_channel", "target_rate", "out_format"):
        assert re.search(r"\b%s\b" % field, header), field


def test_go_sources_are_lexically_balanced_and_import_the_binding_by_module_path():
    for rel in FILES:
        src = _strip(open(os.path.join(GO, rel)).read())
        for a, b in ("()", "[]", "{}"):
            assert src.count(a) == src.count(b), (rel, a, src.count(a), src.count(b))
        assert re.search(r"^package \w+", src, flags=re.M), rel
    for rel in FILES[1:]:
        raw = open(os.path.join(GO, rel)).read()
        assert '"github.com/andrepxx/go-dsp-guitar/gdg"' in raw, rel
        assert not re.search(r'^\s*"gdg"', raw, flags=re.M), rel


def test_overlay_generator_writes_absolute_keys(tmp_path):
    ref = tmp_path / "reference"
    ref.mkdir()
    out = tmp_path / "overlay.json"
    subprocess.run(["sh", os.path.join(GO, "make_overlay.sh"), str(ref), str(out)], check=True, capture_output=True)
    rep = json.load(open(out))["Replace"]
    assert len(rep) == 4
    for key, val in rep.items():
        assert os.path.isabs(key) and key.startswith(str(ref)), key
        assert os.path.isabs(val) and os.path.exists(val), val
    assert any(k.endswith("/gdg/gdg.go") for k in rep)

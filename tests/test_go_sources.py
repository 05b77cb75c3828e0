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


def _split_top_level(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(src, start):
    """text between the parenthesis at src[start] and its partner"""
    depth = 0
    for i in range(start, len(src)):
        if src[i] == "(":
            depth += 1
        elif src[i] == ")":
            depth -= 1
            if depth == 0:
                return src[start + 1:i]
    raise AssertionError("unbalanced call")


def test_every_c_call_of_the_binding_has_the_prototype_s_argument_count():
    """What the cgo type checker would refuse first: a call with the wrong number of arguments.  Prototypes from include/gdg.h, calls from gdg.go
    (comments and strings removed)."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gdg.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(gdg_[a-z0-9_]+)\s*\(", header):
        args = _call_args(header, m.end() - 1).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_top_level(args))
    src = _strip(open(os.path.join(GO, "gdg", "gdg.go")).read())
    calls = 0
    for m in re.finditer(r"\bC\.(gdg_[a-z0-9_]+)\s*\(", src):
        name = m.group(1)
        if name not in protos:
            continue                                   # a type conversion such as C.gdg_batch_input(...) is not a call of a function
        n = len(_split_top_level(_call_args(src, m.end() - 1)))
        assert n == protos[name], "%s called with %d arguments, the header declares %d" % (name, n, protos[name])
        calls += 1
    assert calls >= 40, calls


def test_no_go_file_imports_a_package_it_does_not_use():
    """`imported and not used` is a compile error in Go: every import's name (alias or last path element) must qualify something."""
    for rel in FILES:
        code = _strip(open(os.path.join(GO, rel)).read())
        m = re.search(r"import \(\s*(.*?)\)", code, flags=re.S)
        assert m, rel
        body = code[m.end():]
        raw_block = re.search(r"import \(\s*(.*?)\)", open(os.path.join(GO, rel)).read(), flags=re.S).group(1)
        for line in raw_block.strip().split("\n"):
            mm = re.match(r'\s*(?:(\w+)\s+)?"([^"]+)"', line)
            if not mm:
                continue
            alias = mm.group(1) or mm.group(2).split("/")[-1]
            assert re.search(r"\b%s\." % re.escape(alias), body), "%s imports %s and never uses it" % (rel, mm.group(2))


def test_no_go_function_declares_a_variable_it_never_reads():
    """`declared and not used` is the other compile error a never-compiled Go file tends to carry: every name introduced with := or var inside a
    function must occur at least once more in that function (a lexical check; shadowing could fool it, it has not yet)."""
    for rel in FILES:
        code = _strip(open(os.path.join(GO, rel)).read())
        for m in re.finditer(r"^func [^\n]*\{\s*$", code, flags=re.M):
            start = code.rfind("{", m.start(), m.end())
            depth, j = 0, start
            while True:
                if code[j] == "{":
                    depth += 1
                elif code[j] == "}":
                    depth -= 1
                    if depth == 0:
                        break
                j += 1
            body = code[start:j + 1]
            names = []
            for d in re.finditer(r"(?:^|[\s;{(])((?:\w+\s*,\s*)*\w+)\s*:=", body):
                names += [n.strip() for n in d.group(1).split(",")]
            for d in re.finditer(r"\bvar\s+((?:\w+\s*,\s*)*\w+)\s", body):
                names += [n.strip() for n in d.group(1).split(",")]
            for name in names:
                if name == "_":
                    continue
                assert len(re.findall(r"\b%s\b" % re.escape(name), body)) >= 2, (rel, code[m.start():start].strip()[:80], name)


def test_everything_the_overlays_take_from_the_binding_is_defined_there():
    """gdg.X in an overlay must be an exported function, type, constant or variable of go/gdg/gdg.go, and a method called on a binding object an
    exported method of one of its types (the overlays' other methods belong to the reference's packages and the standard library)."""
    g = _strip(open(os.path.join(GO, "gdg", "gdg.go")).read())
    defined = set(re.findall(r"^func ([A-Z]\w*)\(", g, flags=re.M)) | set(re.findall(r"^type ([A-Z]\w*)\b", g, flags=re.M))
    defined |= set(re.findall(r"^\s*([A-Z]\w*)\s*(?:=|[A-Za-z\[\]*.]+\s*=)", g, flags=re.M))
    assert {"CreateContext", "Context"} <= defined
    for rel in FILES[1:]:
        s = _strip(open(os.path.join(GO, rel)).read())
        used = set(re.findall(r"\bgdg\.([A-Z]\w*)", s))
        assert used, rel
        missing = sorted(u for u in used if u not in defined)
        assert not missing, (rel, missing)


def _binding_signatures():
    """exported functions and methods of gdg.go: name -> [(parameter count, variadic, result count)]"""
    g = _strip(open(os.path.join(GO, "gdg", "gdg.go")).read())
    sigs = {}
    for m in re.finditer(r"^func (?:\(\w+ \*?\w+\) )?([A-Za-z]\w*)\(", g, flags=re.M):
        args = _call_args(g, m.end() - 1)
        parts = [p for p in _split_top_level(args) if p.strip()]
        rest = g[m.end() + len(args) + 1:]
        rest = rest[:rest.index("{")].strip()
        if rest == "":
            n_res = 0
        elif rest.startswith("("):
            n_res = len(_split_top_level(rest[1:rest.rindex(")")]))
        else:
            n_res = 1
        sigs.setdefault(m.group(1), []).append((len(parts), any("..." in p for p in parts), n_res))
    return sigs


def test_calls_of_the_binding_match_its_signatures():
    """Argument counts of every call of a binding function or method in the overlays (and inside the binding), and the number of values on the
    left of an assignment whose right side is such a call, against the definitions in gdg.go -- `not enough arguments in call` and
    `assignment mismatch` are what the compiler would say."""
    sigs = _binding_signatures()
    assert len(sigs) >= 40
    for rel in FILES:
        s = _strip(open(os.path.join(GO, rel)).read())
        if rel != FILES[0]:
            for m in re.finditer(r"\.([A-Z]\w*)\(", s):
                name = m.group(1)
                if name not in sigs:
                    continue
                n = len([p for p in _split_top_level(_call_args(s, m.end() - 1)) if p.strip()])
                assert any(n == c or (var and n >= c - 1) for c, var, _ in sigs[name]), (rel, s.count("\n", 0, m.start()) + 1, name, n, sigs[name])
        for ln, line in enumerate(s.split("\n"), 1):
            m = re.match(r"\s*(?:if\s+)?((?:[\w.\[\]*]+\s*,\s*)*[\w.\[\]*]+)\s*(?::=|=)\s*(?:[\w.\[\]()]+\.)?([A-Za-z]\w*)\((.*)\)\s*(?:;.*\{)?\s*$", line)
            if not m or m.group(2) not in sigs:
                continue
            lhs = len(_split_top_level(m.group(1)))
            assert lhs in {r for _, _, r in sigs[m.group(2)]}, (rel, ln, line.strip())

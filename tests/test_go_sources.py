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
    assert len(rep) == 3                                     # the three REPLACED files; they are pure Go (an overlaid file must not need cgo)
    for key, val in rep.items():
        assert os.path.isabs(key) and key.startswith(str(ref)), key
        assert os.path.isabs(val) and os.path.exists(val), val
        assert 'import "C"' not in open(val).read(), val
    # the cgo binding is NOT an overlay entry: cmd/go runs cgo inside the package directory, so it is copied into the checkout
    assert not any(k.endswith("/gdg/gdg.go") for k in rep)
    copied = ref / "gdg" / "gdg.go"
    assert copied.exists() and copied.read_text() == open(os.path.join(GO, "gdg", "gdg.go")).read()
    # every step of INTEGRATION.md's recipe works on directories that exist on disk
    recipe = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "make_overlay.sh $REF" in recipe and "overlay-only" in recipe
    assert '"<REF>/gdg/gdg.go":' not in recipe


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


# ---- language level and API level: the reference's go.mod says `go 1.16` -------------------------------------------------------

GOLDEN = os.path.join(ROOT, "tests", "golden", "go_exports.json")

# identifiers and syntax that do not exist at -lang=go1.16 (which cmd/go passes for every package of the reference's module,
# the binding's directory included) or in a Go 1.16 standard library
POST_1_16 = [
    (r"\bunsafe\.(Slice|Add|String|StringData|SliceData)\b", "unsafe.Slice/Add are Go 1.17, String/StringData/SliceData 1.20"),
    (r"\bany\b", "`any` is Go 1.18"),
    (r"\bcomparable\b", "`comparable` is Go 1.18"),
    (r"\bfunc\s+\w+\s*\[[^\]]*\]\s*\(", "type parameters are Go 1.18"),
    (r"\btype\s+\w+\s*\[\s*\w+\s+[\w.|~ ]+\]\s*(struct|interface|func|\[|\w)", "generic types are Go 1.18"),
    (r"\batomic\.(Int32|Int64|Uint32|Uint64|Uintptr|Bool|Pointer)\b", "typed atomics are Go 1.19"),
    (r"(?<![.\w])(min|max|clear)\s*\(", "min / max / clear builtins are Go 1.21"),
    (r"\bruntime\.Pinner\b", "runtime.Pinner is Go 1.21"),
    (r"\bfor\s+\w+\s*:=\s*range\s+\d", "range over an integer is Go 1.22"),
    (r"\bfor\s+range\s+\d", "range over an integer is Go 1.22"),
    (r"\b(errors\.Join|strings\.Cut|strings\.CutPrefix|strings\.CutSuffix|strings\.Clone|bytes\.Cut|sync\.OnceFunc|sync\.OnceValue|slices\.|maps\.|cmp\.)",
     "standard-library API newer than Go 1.16"),
    (r"\bmath\.MaxInt\b|\bmath\.MaxUint\b|\bmath\.MinInt\b", "math.MaxInt & co. are Go 1.17"),
    (r"\btime\.(UnixMilli|UnixMicro)\b|\.UnixMilli\(|\.UnixMicro\(", "UnixMilli / UnixMicro are Go 1.17"),
    (r"\bfmt\.Append\w*\(", "fmt.Append* is Go 1.19"),
    (r"\[\s*\.\.\.\s*\]\s*\w+\s*\(\s*\w+\s*\)", "slice-to-array conversion is Go 1.20"),
]

# what the four files may take from the standard library: every selector was in Go 1.16 (checked by hand against the 1.16 API list);
# anything not listed fails the test and has to be looked up before it is added
STDLIB_1_16 = {
    "fmt": {"Errorf", "Sprintf", "Printf", "Sprint"},
    "log": {"Printf", "Println"},
    "math": {"Pow", "Pi", "Abs", "Floor", "Ceil", "Sqrt", "Inf", "IsInf", "IsNaN"},
    "os": {"Getenv"},
    "strconv": {"Atoi", "Itoa", "ParseUint", "ParseInt", "FormatInt"},
    "strings": {"Split", "TrimSpace", "Join"},
    "sync": {"Mutex", "RWMutex", "Once", "WaitGroup", "NewCond", "Cond"},
    "time": {"AfterFunc", "Millisecond", "Now", "Duration", "Second"},
    "unsafe": {"Pointer", "Sizeof"},
}
# methods of standard-library values the files call (sync.Mutex / RWMutex / Cond / Once / WaitGroup, time.Timer / Time, error)
STDLIB_METHODS = {"Lock", "Unlock", "RLock", "RUnlock", "Wait", "Broadcast", "Signal", "Do", "Add", "Done", "Stop", "Before", "After", "Error"}


def _imports(rel):
    raw = open(os.path.join(GO, rel)).read()
    block = re.search(r"import \(\s*(.*?)\)", raw, flags=re.S).group(1)
    out = {}
    for line in block.strip().split("\n"):
        mm = re.match(r'\s*(?:(\w+)\s+)?"([^"]+)"', line)
        if mm:
            out[mm.group(1) or mm.group(2).split("/")[-1]] = mm.group(2)
    return out


def _post_1_16_hits(code):
    return [(why, m.group(0)) for pat, why in POST_1_16 for m in re.finditer(pat, code)]


def test_no_go_file_uses_language_or_library_features_newer_than_the_reference_go_mod():
    """`go 1.16` in the reference's go.mod makes cmd/go compile every package of the module -- the overlaid files and the added gdg directory --
    with -lang=go1.16, and a Go 1.16 toolchain does not know newer library functions at all."""
    golden = json.load(open(GOLDEN))
    assert golden["go"] == "1.16" and golden["module"] == "github.com/andrepxx/go-dsp-guitar"
    for rel in FILES:
        code = _strip(open(os.path.join(GO, rel)).read())
        assert not _post_1_16_hits(code), (rel, _post_1_16_hits(code))
    # the deny-list itself: round 3's Row() (the compile error the review found) and a few siblings must be caught
    for bad in ("in = unsafe.Slice((*float64)(p), frames)", "p = unsafe.Add(p, 8)", "func f(x any) {}", "func Map[T any](x T) T {", "var n atomic.Int64",
                "m := min(a, b)", "var p runtime.Pinner", "for i := range 10 {", "a, b, ok := strings.Cut(s, \",\")"):
        assert _post_1_16_hits(bad), bad
    for good in ("in = (*[1 << 37]float64)(unsafe.Pointer(uintptr(p) + off))[:frames:frames]", "m := this.min(a, b)", "x := math.Max(a, b)"):
        assert not _post_1_16_hits(good), good


def test_standard_library_selectors_are_on_the_go_1_16_allow_list():
    for rel in FILES:
        code = _strip(open(os.path.join(GO, rel)).read())
        for alias, path in _imports(rel).items():
            if "/" in path and path.split("/")[0] == "github.com":
                continue
            assert path in STDLIB_1_16, "%s imports %s: add it to the allow-list after checking the Go 1.16 API" % (rel, path)
            for name in set(re.findall(r"(?<![\w.])%s\.([A-Za-z]\w*)" % re.escape(alias), code)):
                assert name in STDLIB_1_16[path], "%s uses %s.%s: not on the Go 1.16 allow-list" % (rel, path, name)


def test_everything_the_overlays_take_from_the_reference_is_exported_there():
    """effects.X / filter.X / circular.X must be exported by that package of the reference, and an exported method called on any value must belong
    to the binding, to the overlay itself, to one of the reference packages the file imports, or to the standard-library types in use."""
    golden = json.load(open(GOLDEN))
    g = _strip(open(os.path.join(GO, "gdg", "gdg.go")).read())
    binding_methods = set(re.findall(r"^func \(\w+ \*?\w+\) ([A-Z]\w*)\(", g, flags=re.M))
    binding_fields = set(re.findall(r"^\t([A-Z]\w*)(?:\s*,\s*[A-Z]\w*)*\s+[\w\[\]*.]+", g, flags=re.M))
    seen_ref = 0
    for rel in FILES[1:]:
        code = _strip(open(os.path.join(GO, rel)).read())
        own_methods = set(re.findall(r"^func \(\w+ \*?\w+\) ([A-Z]\w*)\(", code, flags=re.M)) | set(re.findall(r"^\t([A-Z]\w*)\(", code, flags=re.M))
        ref_methods, ref_fields = set(), set()
        for alias, path in _imports(rel).items():
            if not path.startswith(golden["module"] + "/") or path.endswith("/gdg"):
                continue
            pkg = path[len(golden["module"]) + 1:]
            assert pkg in golden["packages"], (rel, pkg)
            exp = golden["packages"][pkg]
            names = set(exp["funcs"]) | set(exp["types"]) | set(exp["values"])
            for name in set(re.findall(r"(?<![\w.])%s\.([A-Za-z]\w*)" % re.escape(alias), code)):
                assert name in names, "%s takes %s.%s, which the reference does not export" % (rel, pkg, name)
                seen_ref += 1
            ref_methods |= set(exp["methods"])
            ref_fields |= set(exp["fields"])
        allowed = binding_methods | own_methods | ref_methods | STDLIB_METHODS
        for m in re.finditer(r"(?<=[\w\])])\.([A-Z]\w*)\(", code):
            head = code[:m.start()]
            qual = re.search(r"([A-Za-z_]\w*)$", head)
            if qual and qual.group(1) in _imports(rel):
                continue                                    # pkg.Func(...): checked above
            assert m.group(1) in allowed, "%s line %d calls .%s(): no such method on the binding, the overlay, the reference packages it imports or the standard types" % (
                rel, code.count("\n", 0, m.start()) + 1, m.group(1))
        # exported FIELD reads (p.NumericValue, sh.First ...): a field of an imported reference struct or of a binding struct
        for m in re.finditer(r"(?<=[\w\])])\.([A-Z]\w*)\b(?!\s*\()", code):
            head = code[:m.start()]
            qual = re.search(r"([A-Za-z_]\w*)$", head)
            if qual and qual.group(1) in _imports(rel):
                continue
            assert m.group(1) in ref_fields | binding_fields | binding_methods | own_methods, "%s line %d reads .%s: no such exported field" % (rel, code.count("\n", 0, m.start()) + 1, m.group(1))
    assert seen_ref >= 8, seen_ref


def test_the_overlays_export_everything_their_reference_packages_export():
    """controller, webserver ... keep importing signal / tuner / spatializer: a replacement file has to offer every exported function, type,
    constant and interface method of the file it replaces (each of the three packages is ONE file in the reference)."""
    golden = json.load(open(GOLDEN))
    for rel in FILES[1:]:
        pkg = rel.split("/")[0]
        exp = golden["packages"][pkg]
        code = _strip(open(os.path.join(GO, rel)).read())
        have = set(re.findall(r"^func ([A-Z]\w*)\(", code, flags=re.M)) | set(re.findall(r"^type ([A-Z]\w*)\b", code, flags=re.M))
        have |= set(re.findall(r"^(?:const|var) ([A-Z]\w*)\b", code, flags=re.M))
        for blk in re.finditer(r"^(?:const|var) \((.*?)^\)", code, flags=re.M | re.S):
            have |= set(re.findall(r"^\t([A-Z]\w*)\b", blk.group(1), flags=re.M))
        iface = set(re.findall(r"^\t([A-Z]\w*)\(", code, flags=re.M))
        for name in exp["funcs"] + exp["types"] + exp["values"]:
            assert name in have, "%s does not export %s, the reference's %s does" % (rel, name, pkg)
        for name in exp["methods"]:
            assert name in iface, "%s: interface method %s of the reference's %s is missing" % (rel, name, pkg)


def test_tuner_overlay_keeps_the_reference_s_two_locks():
    """tuner/tuner.go:48-57, :380-412, :582-587: Process takes mutexBuffer only -- no device call on the audio path --, Analyze takes mutexAnalyze and
    holds mutexBuffer (shared) only around Retrieve."""
    code = _strip(open(os.path.join(GO, "tuner", "tuner.go")).read())
    proc = re.search(r"func \(this \*tunerStruct\) Process\(.*?^\}", code, flags=re.S | re.M).group(0)
    assert "mutexBuffer.Lock()" in proc and "Enqueue(" in proc
    assert "ctx" not in proc and "gdg." not in proc and "mutexAnalyze" not in proc
    ana = re.search(r"func \(this \*tunerStruct\) Analyze\(.*?^\}", code, flags=re.S | re.M).group(0)
    assert ana.index("mutexAnalyze.Lock()") < ana.index("mutexBuffer.RLock()") < ana.index("Retrieve(") < ana.index("mutexBuffer.RUnlock()") < ana.index("TunerReplace(") < ana.index("TunerAnalyze(")
    assert "TunerEnqueueStaged(" not in ana            # ONE upload of the whole ring (gdg_tuner_replace refuses any length but NUM_SAMPLES)


# ---- cgo argument TYPES: what the cgo type checker refuses second ---------------------------------------------------------------

def _c_type_to_cgo(ctype):
    """A parameter type of include/gdg.h as cgo spells it in Go ("const double *const *" -> "**C.double")."""
    t = re.sub(r"\bconst\b", " ", ctype)
    t = re.sub(r"\b(restrict|__restrict__)\b", " ", t)
    stars = t.count("*")
    base = " ".join(t.replace("*", " ").split())
    names = {"int": "C.int", "unsigned": "C.uint", "unsigned int": "C.uint", "double": "C.double", "char": "C.char", "long long": "C.longlong",
             "size_t": "C.size_t", "uint32_t": "C.uint32_t", "int32_t": "C.int32_t", "uint8_t": "C.uint8_t", "int8_t": "C.int8_t", "uint64_t": "C.uint64_t"}
    if base == "void":
        assert stars >= 1, ctype
        return "*" * (stars - 1) + "unsafe.Pointer"
    if base in names:
        return "*" * stars + names[base]
    assert re.fullmatch(r"gdg_[a-z0-9_]+", base), "unknown C type %r" % ctype
    return "*" * stars + "C." + base


def _header_prototypes():
    """name -> (return type as cgo, [parameter types as cgo]); struct name -> {field: cgo type}"""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gdg.h")).read(), flags=re.S)
    protos, structs = {}, {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(gdg_[a-z0-9_]+)\s*;", header, flags=re.S):
        fields = {}
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *more = [d.strip() for d in decl.split(",")]
            mm = re.match(r"(.*?)(\**)\s*([A-Za-z_][A-Za-z0-9_]*)$", first)
            base = mm.group(1).strip()
            fields[mm.group(3)] = _c_type_to_cgo(base + mm.group(2))
            for extra in more:
                mm2 = re.match(r"(\**)\s*([A-Za-z_][A-Za-z0-9_]*)$", extra)
                fields[mm2.group(2)] = _c_type_to_cgo(base + mm2.group(1))
        structs[m.group(2)] = fields
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*gdg_[a-z0-9_]+\s*;", "", header, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(gdg_[a-z0-9_]+)\s*\(", body):
        ret = m.group(1).strip()
        if not ret or ret.startswith("#") or "return" in ret or "typedef" in ret:
            continue
        args = _call_args(body, m.end() - 1).strip()
        params = []
        if args not in ("", "void"):
            for a in _split_top_level(args):
                a = re.sub(r"\s*=\s*[^,]+$", "", a.strip())
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)?$", a)
                typ = mm.group(1).strip() if (mm.group(2) and mm.group(1).strip()) else a
                params.append(_c_type_to_cgo(typ))
        protos[m.group(2)] = (None if ret == "void" else _c_type_to_cgo(ret), params)
    return protos, structs


def _go_functions(src):
    """(name, body text, parameter text) of every function of a stripped Go source"""
    out = []
    for m in re.finditer(r"^func\s*(\([^)]*\))?\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M):
        params = _call_args(src, m.end() - 1)
        brace = src.index("{", m.end() - 1 + len(params) + 1)
        depth = 0
        for i in range(brace, len(src)):
            if src[i] == "{":
                depth += 1
            elif src[i] == "}":
                depth -= 1
                if depth == 0:
                    out.append((m.group(2), src[brace:i + 1], params, brace))
                    break
    return out


_CAST = r"(?:C\.[A-Za-z0-9_]+|unsafe\.Pointer)"


def _type_of_expr(expr, body, upto, structs, ctx_fields):
    """cgo type of an argument expression, or None for an untyped constant / nil (fits any numeric / any pointer)."""
    e = expr.strip()
    if e == "nil":
        return "nil"
    if re.fullmatch(r"-?\d+", e):
        return "const"
    m = re.match(r"\((\*+)(%s)\)\s*\(" % _CAST, e)
    if m:
        return m.group(1) + m.group(2)
    m = re.match(r"(%s)\s*\(" % _CAST, e)
    if m:
        return m.group(1)
    m = re.match(r"([a-z][A-Za-z0-9_]*)\s*\(", e)                       # a Go helper of the file (cbool): its single result
    if m and m.group(1) in _GO_RESULTS and len(_GO_RESULTS[m.group(1)]) == 1:
        return _GO_RESULTS[m.group(1)][0]
    if re.fullmatch(r"(this|c)\.ctx", e):
        return "*C.gdg_ctx"
    if re.fullmatch(r"&(this|c)\.ctx", e):
        return "**C.gdg_ctx"
    m = re.fullmatch(r"(this|c)\.([a-z_]+)", e)
    if m and m.group(2) in ctx_fields:
        return ctx_fields[m.group(2)]
    amp = e.startswith("&")
    core = e[1:] if amp else e
    m = re.fullmatch(r"([A-Za-z_][A-Za-z0-9_]*)(\[[^\]]*\])?(?:\.([a-z_]+))?", core)
    if not m:
        return "?" + e
    name, index, field = m.group(1), m.group(2), m.group(3)
    t = _declared_type(name, body[:upto])
    if t is None:
        return "?" + e
    if index:
        if not t.startswith("[]"):
            return "?" + e
        t = t[2:]
    if field:
        sname = t.replace("C.", "")
        if sname not in structs or field not in structs[sname]:
            return "?" + e
        t = structs[sname][field]
    return ("*" + t) if amp else t


_GO_RESULTS = {}


def _go_results(src):
    """Go function name -> list of result types (from `func name(...) (T1, T2)` / `func name(...) T`)"""
    out = {}
    for m in re.finditer(r"^func\s*(?:\([^)]*\))?\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M):
        params = _call_args(src, m.end() - 1)
        rest = src[m.end() + len(params) + 1:]
        rest = rest[:rest.index("{")].strip()

        def norm(r):
            r = r.strip()
            mm = re.match(r"[A-Za-z_][A-Za-z0-9_]*\s+(.*)$", r)            # a named result: drop the name
            if mm and not r.startswith("func"):
                r = mm.group(1).strip()
            return re.sub(r"^\*\[[^\]]*\]", "[]", r)                      # pointer to an array: indexes like a slice
        if rest.startswith("("):
            out[m.group(1)] = [norm(r) for r in _split_top_level(rest[1:rest.rindex(")")])]
        elif rest:
            out[m.group(1)] = [norm(rest)]
    return out


def _declared_type(name, before):
    """type of the local `name` from its LAST declaration in the text before its use"""
    best, pos = None, -1
    for mm in re.finditer(r"((?:[A-Za-z_][A-Za-z0-9_]*\s*,\s*)*[A-Za-z_][A-Za-z0-9_]*)\s*:=\s*([a-z][A-Za-z0-9_]*)\s*\(", before):
        lhs = [x.strip() for x in mm.group(1).split(",")]
        if name in lhs and mm.group(2) in _GO_RESULTS and len(_GO_RESULTS[mm.group(2)]) == len(lhs):
            t = _GO_RESULTS[mm.group(2)][lhs.index(name)]
            if re.fullmatch(r"\**(?:\[\])?\**%s" % _CAST, t) and mm.start() > pos:
                best, pos = t, mm.start()
    m = None
    for m in re.finditer(r"\b%s\s*:=\s*C\.CString\s*\(" % re.escape(name), before):
        pass
    if m:
        best, pos = "*C.char", m.start()                # C.CString returns *C.char
    for mm in re.finditer(r"((?:[A-Za-z_][A-Za-z0-9_]*\s*,\s*)+[A-Za-z_][A-Za-z0-9_]*)\s*:=\s*([^\n]+)", before):     # a, b := X, Y
        lhs = [x.strip() for x in mm.group(1).split(",")]
        rhs = _split_top_level(mm.group(2))
        if name in lhs and len(rhs) == len(lhs) and mm.start() > pos:
            r = rhs[lhs.index(name)].strip()
            if re.match(r"C\.(malloc|calloc|CBytes)\s*\(", r):
                best, pos = "unsafe.Pointer", mm.start()
    for mm in re.finditer(r"\b%s\s*:=\s*C\.(malloc|calloc|CBytes)\s*\(" % re.escape(name), before):
        if mm.start() > pos:
            best, pos = "unsafe.Pointer", mm.start()
    pats = [
        (r"\bvar\s+(?:[A-Za-z_][A-Za-z0-9_]*\s*,\s*)*%s\b(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*\s+(\**(?:\[\])?\**%s)" % (re.escape(name), _CAST), lambda m: m.group(1)),
        (r"\b%s\s*:=\s*make\(\s*(\[\]\**%s)" % (re.escape(name), _CAST), lambda m: m.group(1)),
        (r"\b%s\s*:=\s*(%s)\s*\{" % (re.escape(name), _CAST), lambda m: m.group(1)),
        (r"\b%s\s*:=\s*(C\.(?!gdg_|CString\b)[A-Za-z0-9_]+|unsafe\.Pointer)\s*\(" % re.escape(name), lambda m: m.group(1)),
        (r"\b%s\s*:=\s*\((\*+%s)\)\s*\(" % (re.escape(name), _CAST), lambda m: m.group(1)),
        (r"\b%s\s*:=\s*\(\*\[[^\]]*\](\**%s)\)\s*\(" % (re.escape(name), _CAST), lambda m: "[]" + m.group(1)),      # pointer to an array: indexes like a slice
        (r"\b%s\s*:=\s*&([A-Za-z_][A-Za-z0-9_]*)\b" % re.escape(name), None),
        (r"\b%s\s*:=\s*(C\.gdg_[a-z0-9_]+)\s*\(" % re.escape(name), "call"),
    ]
    for pat, fn in pats:
        for m in re.finditer(pat, before):
            if m.start() > pos:
                if fn is None:
                    inner = _declared_type(m.group(1), before[:m.start()])
                    if inner is None:
                        continue
                    best, pos = "*" + inner, m.start()
                elif fn == "call":
                    best, pos = ("ret:" + m.group(1)[2:]), m.start()
                else:
                    best, pos = fn(m), m.start()
    return best


def test_every_c_call_of_the_binding_passes_the_prototype_s_types():
    """cgo maps every C type to ONE Go type and converts nothing implicitly: C.int where the header says size_t, *C.int where it says
    const int32_t *, a Go int where it says int -- each is a compile error.  So: the cgo type of every argument of every C.gdg_* call, inferred
    from its conversion (C.int(x), (*C.double)(p), unsafe.Pointer(p)), from `nil` / an untyped constant, or from the declaration of the local it
    names (var x C.int, make([]C.gdg_batch_input, n), &arr[0], o.out_format ...), against the prototype; and every result against its use
    (this.err takes C.int, C.GoString takes *C.char).  An argument the test cannot type is a failure, not a pass."""
    protos, structs = _header_prototypes()
    raw = open(os.path.join(GO, "gdg", "gdg.go")).read()
    src = _strip(raw)
    _GO_RESULTS.clear()
    _GO_RESULTS.update(_go_results(src))
    # fields of the Go struct `Context` that hold C values
    ctx_fields = {}
    m = re.search(r"type Context struct \{(.*?)\n\}", src, flags=re.S)
    for line in m.group(1).splitlines():
        mm = re.match(r"\s*([a-z_, ]+?)\s+(\**(?:%s|unsafe\.Pointer))\s*$" % _CAST, line)
        if mm:
            for nm in mm.group(1).split(","):
                ctx_fields[nm.strip()] = mm.group(2)
    # the helper every status goes through
    m = re.search(r"func \(this \*Context\) err\(rc (C\.[a-z0-9_]+)\)", src)
    assert m and m.group(1) == "C.int", "Context.err must take C.int: every gdg_* status is an int"
    checked = 0
    for fname, body, params, _ in _go_functions(src):
        for m in re.finditer(r"\bC\.(gdg_[a-z0-9_]+)\s*\(", body):
            name = m.group(1)
            if name not in protos:
                continue                                   # C.gdg_batch_input{...} / a conversion
            ret, want = protos[name]
            args = _split_top_level(_call_args(body, m.end() - 1))
            assert len(args) == len(want), (fname, name)
            for k, (a, w) in enumerate(zip(args, want)):
                got = _type_of_expr(a, body, m.start(), structs, ctx_fields)
                where = "%s: argument %d of %s (%s)" % (fname, k + 1, name, a.strip())
                assert not got.startswith("?"), "cannot type " + where
                if got.startswith("ret:"):
                    got = protos[got[4:]][0]
                if got == "nil":
                    assert w.startswith("*") or w == "unsafe.Pointer", where + ": nil for " + w
                elif got == "const":
                    assert not w.startswith("*") and w != "unsafe.Pointer", where + ": a constant for " + w
                else:
                    assert got == w, where + ": %s, the header wants %s" % (got, w)
                checked += 1
            # the result's use
            pre = body[max(0, m.start() - 40):m.start()]
            if re.search(r"\.err\(\s*$", pre):
                assert ret == "C.int", (fname, name, ret)
            if re.search(r"C\.GoString\(\s*$", pre):
                assert ret == "*C.char", (fname, name, ret)
            if re.search(r"\bint\(\s*$", pre):
                assert ret in ("C.int", "C.size_t"), (fname, name, ret)
    assert checked > 150, checked

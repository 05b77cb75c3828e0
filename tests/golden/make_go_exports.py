"""Extracts, from the reference tree, the EXPORTED names of the packages the Go overlays import -- what `go build` would resolve
`effects.X`, `filter.X`, `circular.X` and method calls on their interface values against -- plus the module's language level.

Run in the authoring container, where /root/reference is readable:   python tests/golden/make_go_exports.py
Writes tests/golden/go_exports.json (names only: data, not source).  tests/test_go_sources.py checks the overlays against it, so the
check travels without the reference tree.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
PACKAGES = ["effects", "filter", "circular", "fft", "resample", "wave", "level", "metronome", "random", "oversampling", "signal", "tuner", "spatializer"]


def strip(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(\\.|[^"\\\n])*"', '""', src)
    src = re.sub(r"`[^`]*`", '""', src)
    return src


def exports_of(pkg_dir):
    funcs, types, values, methods, fields = set(), set(), set(), set(), set()
    for name in sorted(os.listdir(pkg_dir)):
        if not name.endswith(".go") or name.endswith("_test.go"):
            continue
        src = strip(open(os.path.join(pkg_dir, name)).read())
        funcs |= set(re.findall(r"^func ([A-Z]\w*)\(", src, flags=re.M))
        methods |= set(re.findall(r"^func \(\w+ \*?\w+\) ([A-Z]\w*)\(", src, flags=re.M))
        types |= set(re.findall(r"^type ([A-Z]\w*)\b", src, flags=re.M))
        # const / var: single declarations and parenthesised blocks
        values |= set(re.findall(r"^(?:const|var) ([A-Z]\w*)\b", src, flags=re.M))
        for blk in re.finditer(r"^(?:const|var) \((.*?)^\)", src, flags=re.M | re.S):
            values |= set(re.findall(r"^\t([A-Z]\w*)\b", blk.group(1), flags=re.M))
        # interface methods and exported struct fields of exported types
        for m in re.finditer(r"^type ([A-Z]\w*) (interface|struct) \{(.*?)^\}", src, flags=re.M | re.S):
            body = m.group(3)
            if m.group(2) == "interface":
                methods |= set(re.findall(r"^\t([A-Z]\w*)\(", body, flags=re.M))
            else:
                fields |= set(re.findall(r"^\t([A-Z]\w*)\s", body, flags=re.M))
    return {"funcs": sorted(funcs), "types": sorted(types), "values": sorted(values), "methods": sorted(methods), "fields": sorted(fields)}


def main():
    gomod = open(os.path.join(REF, "go.mod")).read()
    out = {
        "module": re.search(r"^module (\S+)", gomod, flags=re.M).group(1),
        "go": re.search(r"^go (\S+)", gomod, flags=re.M).group(1),
        "packages": {p: exports_of(os.path.join(REF, p)) for p in PACKAGES if os.path.isdir(os.path.join(REF, p))},
    }
    path = os.path.join(HERE, "go_exports.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, {p: sum(len(v) for v in e.values()) for p, e in out["packages"].items()})


if __name__ == "__main__":
    main()

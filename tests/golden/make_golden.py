#!/usr/bin/env python3
"""Extract the literal golden vectors held by the reference's own unit tests into JSON fixtures.

Run ONLY in the authoring container (needs /root/reference); the fixtures it writes are
committed, the reference never travels.  Only DATA is extracted: the float / complex / uint
literal tables of each `func TestXxx` in the listed *_test.go files, keyed by test function
and variable name, with their file:line provenance.  No reference code is copied.

    python tests/golden/make_golden.py
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

FILES = {
    "fft": "fft/fft_test.go",
    "oversampling": "oversampling/oversampling_test.go",
    "resample": "resample/resample_test.go",
    "random": "random/random_test.go",
    "circular": "circular/circular_test.go",
    "wave": "wave/wave_test.go",
    "level": "level/level_test.go",
}

NUM = r"-?(?:0x[0-9a-fA-F]+|\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+)"


def parse_number(tok):
    tok = tok.strip()
    if tok.lower().startswith("0x"):
        return int(tok, 16)
    if re.fullmatch(r"-?\d+", tok):
        return int(tok)
    return float(tok)


def find_matching(src, start):
    """index of the brace matching src[start] == '{'"""
    depth = 0
    for i in range(start, len(src)):
        if src[i] == "{":
            depth += 1
        elif src[i] == "}":
            depth -= 1
            if depth == 0:
                return i
    raise ValueError("unbalanced braces")


def parse_flat(body, elem_type):
    if elem_type == "complex128":
        return [[float(a), float(b)] for a, b in re.findall(r"complex\(\s*(%s)\s*,\s*(%s)\s*\)" % (NUM, NUM), body)]
    return [parse_number(t) for t in re.findall(NUM, body)]


def parse_literal(src, pos, type_str):
    """src[pos] is the '{' opening a literal of Go type type_str ('[]T' or '[][]T')."""
    end = find_matching(src, pos)
    body = src[pos + 1:end]
    if type_str.startswith("[][]"):
        elem = type_str[4:]
        rows = []
        for m in re.finditer(r"\[\]%s\s*\{" % re.escape(elem), body):
            b0 = m.end() - 1
            b1 = find_matching(body, b0)
            rows.append(parse_flat(body[b0 + 1:b1], elem))
        return rows, end
    return parse_flat(body, type_str[2:]), end


def extract(path):
    src = open(os.path.join(REF, path)).read()
    result = {}
    for fm in re.finditer(r"^func (Test\w+)\(t \*testing\.T\) \{", src, re.M):
        name = fm.group(1)
        f_end = find_matching(src, fm.end() - 1)
        fbody_start = fm.end()
        entry = {}
        for vm in re.finditer(r"(\w+) := ((?:\[\])+(?:float64|complex128|uint64|int|uint32|byte|int32))\s*\{", src[fbody_start:f_end]):
            var, type_str = vm.group(1), vm.group(2)
            pos = fbody_start + vm.end() - 1
            value, _ = parse_literal(src, pos, type_str)
            line = src.count("\n", 0, fbody_start + vm.start()) + 1
            entry[var] = {"type": type_str, "line": line, "value": value}
        if entry:
            result[name] = entry
    return result


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not mounted; fixtures are already committed")
    for key, path in FILES.items():
        data = {"source": path, "tests": extract(path)}
        with open(os.path.join(OUT, key + ".json"), "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
        print(key, {k: sorted(v) for k, v in data["tests"].items()})


if __name__ == "__main__":
    main()

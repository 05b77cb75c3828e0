#!/usr/bin/env python3
"""Extract the parameter tables (name, type, unit, min, max, default, discrete values) of the 21
effects units from the reference's create*() functions into tests/golden/params.json.
DATA only; run in the authoring container (needs /root/reference)."""
import glob
import json
import os
import re

REF = "/root/reference/effects"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "params.json")
UNIT_CONSTS = ["UNIT_SIGNALGENERATOR", "UNIT_NOISEGATE", "UNIT_BANDPASS", "UNIT_AUTOWAH", "UNIT_AUTOYOY", "UNIT_COMPRESSOR",
               "UNIT_OCTAVER", "UNIT_EXCESS", "UNIT_FUZZ", "UNIT_OVERDRIVE", "UNIT_DISTORTION", "UNIT_TONESTACK", "UNIT_CHORUS",
               "UNIT_FLANGER", "UNIT_PHASER", "UNIT_TREMOLO", "UNIT_RINGMODULATOR", "UNIT_DELAY", "UNIT_REVERB", "UNIT_POWERAMP",
               "UNIT_CABINET"]


def field(block, name, pattern):
    m = re.search(name + r":\s*" + pattern, block)
    return m.group(1) if m else None


def main():
    table = {}
    for path in sorted(glob.glob(os.path.join(REF, "*.go"))):
        src = open(path).read()
        m = re.search(r"func create\w+\(\) Unit \{(.*)", src, re.S)
        if not m:
            continue
        body = m.group(1)
        unit = re.search(r"unitType:\s*(\w+)", body).group(1)
        line0 = src[:m.start(1)].count("\n") + 1
        params = []
        for pm in re.finditer(r"Parameter\{(.*?)\n\t\t\t\t\},", body, re.S):
            blk = pm.group(1)
            dv = re.search(r"DiscreteValues:\s*\[\]string\{(.*?)\}", blk, re.S)
            params.append({
                "Name": field(blk, "Name", r'"([^"]*)"'),
                "Type": field(blk, "Type", r"(\w+)"),
                "PhysicalUnit": field(blk, "PhysicalUnit", r'"([^"]*)"'),
                "Minimum": int(field(blk, "Minimum", r"(-?\d+)")),
                "Maximum": int(field(blk, "Maximum", r"(-?\d+)")),
                "NumericValue": int(field(blk, "NumericValue", r"(-?\d+)")),
                "DiscreteValueIndex": int(field(blk, "DiscreteValueIndex", r"(-?\d+)")),
                "DiscreteValues": re.findall(r'"([^"]*)"', dv.group(1)) if dv else [],
            })
        table[str(UNIT_CONSTS.index(unit))] = {"unit": unit, "source": "effects/%s:%d" % (os.path.basename(path), line0), "params": params}
    json.dump(table, open(OUT, "w"), indent=1, sort_keys=True, ensure_ascii=False)
    print({k: len(v["params"]) for k, v in table.items()})


if __name__ == "__main__":
    main()

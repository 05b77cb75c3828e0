#!/usr/bin/env python3
"""The reference's shipped impulse-response library as a test fixture: DATA only -- the 16-bit sample words of the 24 mono 96 kHz WAV
files under ir/ and the name / gain compensation of each from ir/index.json.

    python tests/golden/make_ir_golden.py       # needs /root/reference; rewrites tests/golden/ir_library.npz (~170 KB)

filter.Import (filter/filter.go:704-790) decodes each file (wave: sample = word * 2 / 65535), resamples it to the seven supported
rates with resample.Time and keeps 10^(compensation / 20) as its gain compensation; tests/test_ir_library.py walks the same path."""
import json
import os
import wave

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

with open(os.path.join(REF, "ir", "index.json")) as f:
    index = json.load(f)
arrays, names, comps, paths = {}, [], [], []
for i, d in enumerate(index):
    w = wave.open(os.path.join(REF, d["Path"]))
    assert (w.getnchannels(), w.getframerate(), w.getsampwidth()) == (1, 96000, 2), d
    arrays["ir%02d" % i] = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    w.close()
    names.append(d["Name"]); comps.append(int(d["Compensation"])); paths.append(d["Path"])
np.savez_compressed(os.path.join(HERE, "ir_library.npz"), names=np.array(names), compensation=np.array(comps, dtype=np.int32),
                    paths=np.array(paths), sample_rate=np.int32(96000), **arrays)
print("wrote %d impulse responses, %d samples" % (len(names), sum(a.size for a in arrays.values())))

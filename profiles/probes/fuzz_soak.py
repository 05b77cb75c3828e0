#!/usr/bin/env python3
"""More seeds of the randomised parity tests than the suite runs (tests/test_gpu_fuzz.py), one after the other; prints the failures.
    python profiles/probes/fuzz_soak.py [first_seed [count]]"""
import os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import __graft_entry__ as entry
import test_gpu_fuzz as F
oracle = entry.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = [n for n in dir(F) if n.startswith("test_random_")]
bad = 0
for name in names:
    fn = getattr(F, name)
    ok = 0
    for seed in range(first, first + count):
        try:
            fn(oracle, seed) if fn.__code__.co_argcount == 2 else fn(seed)
            ok += 1
        except Exception as e:                                  # noqa: BLE001
            bad += 1
            print("FAIL %s[%d]: %s" % (name, seed, str(e).split("\n")[0][:300]), flush=True)
    print("%-62s %d / %d seeds pass" % (name, ok, count), flush=True)
print("failures:", bad)

"""Soak of tests/test_gpu_segf.py::test_random_in_place_chains_on_the_two_per_cu_kernel over further seeds (not part of the suite):
    python profiles/probes/soak_segf.py [first seed] [last seed]"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import pytest
import test_gpu_segf as t
import __graft_entry__ as e
orc = e.load_oracle(); orc.build()
class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 424)
for seed in range(lo, hi):
    try:
        t.test_random_in_place_chains_on_the_two_per_cu_kernel(orc, MP(), seed)
    except AssertionError as ex:
        bad += 1
        print("seed", seed, "FAILED", str(ex)[:300], flush=True)
print("soak done, failures:", bad)

"""bench.py puts HBM counter traffic into its JSON line from profiles/pmc_fir_mac.json (the counters cannot be sampled from inside the
process).  Round 4's line carried a segment-kernel figure from before the last kernel change.  These tests keep the file honest:
it is GENERATED from a committed rocprofv3 summary (profiles/make_pmc_json.py), and that summary must not be older -- in git history -- than
the kernel sources it describes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def test_pmc_json_is_what_the_generator_makes_from_its_source():
    import make_pmc_json
    have = json.load(open(os.path.join(ROOT, "profiles", "pmc_fir_mac.json")))
    assert os.path.exists(os.path.join(ROOT, have["source"])), have["source"]
    assert have == make_pmc_json.build(have["source"]), "run: python profiles/make_pmc_json.py %s" % have["source"]
    # the figures bench.py reads are there and plausible: counter traffic of the roofline kernel within 10 % of its algorithmic bytes
    alg = 512 * (2 * 8 * 8192 * 16 + 8192 * 8)
    assert 0.95 * alg <= have["traffic_bytes_per_launch"] <= 1.10 * alg
    assert have["segment_kernel"]["traffic_bytes_per_launch"] > 67e6


def _last_commit_time(paths):
    out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%ct", "--"] + paths, capture_output=True, text=True)
    return int(out.stdout.strip()) if out.returncode == 0 and out.stdout.strip() else None


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, ".git")), reason="no git history here (the GPU box gets a snapshot)")
def test_the_summary_is_not_older_than_the_kernels_it_describes():
    """The roofline kernel lives in fir.hip, the segment kernel in seg.hip (both builds), both are shaped by the Makefile's flags: a commit that
    touches any of them after the summary was committed means the summary describes another build -- re-run profiles/run_rocprof.sh, commit
    the new summary, regenerate the json."""
    have = json.load(open(os.path.join(ROOT, "profiles", "pmc_fir_mac.json")))
    t_summary = _last_commit_time([have["source"]])
    if t_summary is None:
        pytest.skip("the summary is not committed yet")
    csrc = "go-dsp-guitar_amd/csrc/"
    t_kernels = _last_commit_time([csrc + "fir.hip", csrc + "seg.hip", csrc + "Makefile"])
    assert t_kernels is not None
    assert t_summary >= t_kernels, ("%s was committed before the last change of fir.hip / seg.hip / Makefile: profile the current build "
                                    "(bash profiles/run_rocprof.sh <tag>) and regenerate profiles/pmc_fir_mac.json" % have["source"])

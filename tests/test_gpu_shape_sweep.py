"""The launch-shape thresholds on chains the bench never runs (round-5 review, item 4): for three chains and eight channel counts the library's
default is timed against every single decision flipped to the other side of its threshold (profiles/probes/shape_sweep.py; the table of the
full run: profiles/shape_sweep_r06.txt).  The test fails when the default is more than 5 % slower than an alternative in a cell that is not a
documented trade-off -- measured three times, so one noisy sample cannot fail it.

Documented trade-offs (a different chain wants the opposite side of the SAME threshold, and windows and per-frame calls must take the same
kernel to give the same bits):
  * seg_two_per_cu_min_channels at 96 channels: the bench's chain loses 13 % in per-frame calls with the two-per-CU kernel there (round 5),
    the no-reverb chain gains 7 % in windows."""
import importlib.util
import os

import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KNOWN_TRADE_OFFS = {("a_no_reverb", 96, "window", "seg_two_per_cu_min_channels")}


def test_the_default_shape_is_never_much_slower_than_the_other_side_of_a_threshold():
    spec = importlib.util.spec_from_file_location("shape_sweep", os.path.join(ROOT, "profiles", "probes", "shape_sweep.py"))
    sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sweep)
    pkg = entry.load_package()
    lines = []
    rows, bad = sweep.sweep(pkg, sorted(sweep.CHAINS), [32, 64, 96, 128, 192, 256, 448, 512], log=lines.append)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "shape_sweep_from_pytest.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    bad = [r for r in bad if (r[0], r[1], r[2], r[3]) not in KNOWN_TRADE_OFFS]
    assert len(rows) >= 120
    assert not bad, "default more than 5 %% slower than the flipped shape:\n" + "\n".join(
        "%s %d ch %s: %s = %d: default %.1f us, flipped %.1f us (x %.3f)" % r for r in bad)

"""The launch-shape thresholds on chains the bench never runs (round-5 review, item 4): for three chains and eight channel counts the library's
default is timed against every single decision flipped to the other side of its threshold (profiles/probes/shape_sweep.py; the table of the
full run: profiles/shape_sweep_r06.txt).  The test fails when the default is more than 5 % slower than an alternative in a cell that is not a
documented trade-off -- measured three times, so one noisy sample cannot fail it.

One trade-off is built in: seg_two_per_cu_min_channels serves windows AND per-frame calls (both must take the same kernel to give the same
bits).  A cell where one call kind would like the other kernel is excused when the OTHER call kind of the same chain and
channel count is faster with the default -- e.g. 96 channels without a reverb: windows would gain 7 % from the two-per-CU kernel, per-frame
calls lose 4 % (and the bench's chain 13 %, round 5); 256 channels at 96 kHz with an oversampled overdrive: per-frame calls would gain 5 % from
the general kernel, windows lose 5 %."""
import importlib.util
import os

import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    ratio = {(r[0], r[1], r[2], r[3]): r[7] for r in rows}

    def excused(r):
        if r[3] != "seg_two_per_cu_min_channels":
            return False
        other = ratio.get((r[0], r[1], "window" if r[2] == "frame" else "frame", r[3]))
        return other is not None and other < 1.0            # the other call kind is faster with the default: one threshold, two wishes

    bad = [r for r in bad if not excused(r)]
    assert len(rows) >= 120
    assert not bad, "default more than 5 %% slower than the flipped shape:\n" + "\n".join(
        "%s %d ch %s: %s = %d: default %.1f us, flipped %.1f us (x %.3f)" % r for r in bad)

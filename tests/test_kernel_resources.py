"""Register and scratch budgets of the kernels whose speed depends on them, read from the compiler's own summary (no GPU needed: hipcc
cross-compiles).  Two workgroups per CU need <= 128 vector registers per lane -- and the two-per-CU segment kernel ran 15 % slower at 128 than at
120; a `noinline` unit in that build costs 20-36 callee-saved register saves per lane and call, a third of the kernel's HBM
traffic (profiles/experiments/README.md, r04): the units are inlined (both builds; the general one keeps calls for the oversampled shapers and the any-size all-pass)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "go-dsp-guitar_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-x", "hip"]


def summary(tmp_path, source, extra):
    out = str(tmp_path / (source + ".s"))
    subprocess.run([HIPCC] + FLAGS + extra + [os.path.join(CSRC, source), "-o", out], check=True, timeout=900, cwd=CSRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @.*?^; NumVgprs: (\d+).*?^; ScratchSize: (\d+)", text, re.S | re.M):
        res[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return res


SEG_FLAGS = ["-ffp-contract=off", "-mllvm", "-disable-machine-licm"]          # the Makefile's seg.o / segf.o rules


def test_makefile_compiles_the_segment_kernels_with_these_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "SEG_LICM := -mllvm -disable-machine-licm" in mk
    assert mk.count("-ffp-contract=off $(SEG_LICM)") == 3          # seg.o, segf.o (-DSEG_FAST), segt.o (-DSEG_TILE)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_two_per_cu_segment_kernel_keeps_120_registers_and_has_no_unit_calls(tmp_path):
    res = summary(tmp_path, "seg.hip", SEG_FLAGS + ["-DSEG_FAST"])
    kernels = {k: v for k, v in res.items() if "segf_kernel" in k}
    assert len(kernels) == 3, sorted(res)            # one frame per launch, the walk, a workgroup per frame (WAVE)
    for name, (vgprs, scratch) in kernels.items():
        assert vgprs <= 120, (name, vgprs)
        assert scratch <= 64, (name, scratch)
    assert [k for k in res if "kernel" not in k] == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_general_segment_kernel_calls_only_the_oversampled_units(tmp_path):
    res = summary(tmp_path, "seg.hip", SEG_FLAGS)
    kernels = {k: v for k, v in res.items() if "seg_kernel" in k}
    assert len(kernels) == 3, sorted(res)
    for name, (vgprs, scratch) in kernels.items():
        assert vgprs <= 128, (name, vgprs)                 # 1024 threads per workgroup
        assert scratch <= 128, (name, scratch)
    callees = sorted(k for k in res if "kernel" not in k)
    assert len(callees) == 3 and all(any(n in k for n in ("unit_shaper", "unit_fuzz_os", "allpass_generic")) for k in callees), callees


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_tile_segment_kernel_fits_512_threads_without_scratch(tmp_path):
    """seg.hip -DSEG_TILE: a channel's frame on two workgroups of 512 threads (2 waves per SIMD: 256 registers per lane), every unit inlined"""
    res = summary(tmp_path, "seg.hip", SEG_FLAGS + ["-DSEG_TILE"])
    kernels = {k: v for k, v in res.items() if "segt_kernel" in k}
    assert len(kernels) == 1, sorted(res)
    for name, (vgprs, scratch) in kernels.items():
        assert vgprs <= 256 and scratch == 0, (name, vgprs, scratch)
    callees = sorted(k for k in res if "kernel" not in k)
    assert all("allpass_generic" in k for k in callees), callees        # (the extra workgroups' reverb at rates far above 192 kHz)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_one_buffer_transforms_fit_two_workgroups_per_cu(tmp_path):
    res = summary(tmp_path, "fir.hip", [])
    for key in ("fir_inv13h_kernel", "fir_fwd13wh_kernel"):
        found = {k: v for k, v in res.items() if key in k}
        assert found, (key, sorted(res)[:5])
        for name, (vgprs, scratch) in found.items():
            assert vgprs <= 128 and scratch == 0, (name, vgprs, scratch)

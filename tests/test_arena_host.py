"""csrc/arena.h on the host: the sub-allocator behind every unit's device state, run over plain memory (tests/native/arena_check.cpp).
No GPU: the book-keeping is a template over a four-call backend, and the HIP backend in ctx.h adds nothing to it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def arena_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("arena") / "arena_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I", os.path.join(ROOT, "go-dsp-guitar_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "arena_check.cpp"), "-o", exe], check=True, timeout=300)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_arena_churn_on_the_host(arena_check, seed):
    """Zeros where zeros are promised (round 4 found a hole that straddled the never-used mark handing out used space as zeros), no overlap,
    page alignment of large blocks, at most one spare chunk, nothing leaked."""
    r = subprocess.run([arena_check, str(seed), "12000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("seed", [5, 6])
def test_deferred_trimming_frees_only_in_trim(arena_check, seed):
    """The device arena's mode: freeing a chunk waits for the whole device, so release() -- reachable from a process call through a
    parameter patch -- must never do it; trim() does, where the caller has drained the stream anyway."""
    r = subprocess.run([arena_check, str(seed), "12000", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr

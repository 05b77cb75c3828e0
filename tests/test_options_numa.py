"""gdg_ctx_set_option's key table and the sysfs side of option "numa" (include/gdg.h) -- the parts that need no device."""
import os

import pytest

import __graft_entry__ as entry


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


def make_sysfs(root, pci, node, cpulist):
    d = root / "bus" / "pci" / "devices" / pci
    d.mkdir(parents=True)
    (d / "numa_node").write_text("%d\n" % node)
    if node >= 0:
        n = root / "devices" / "system" / "node" / ("node%d" % node)
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")


def test_numa_probe_reads_node_and_cpu_list(pkg, tmp_path):
    make_sysfs(tmp_path, "0000:05:00.0", 1, "64-127,192-255")
    node, cpus = pkg.numa_probe(str(tmp_path), "0000:05:00.0")
    assert node == 1 and cpus == list(range(64, 128)) + list(range(192, 256))
    # hipDeviceGetPCIBusId prints upper-case hex; sysfs names are lower case
    make_sysfs(tmp_path, "0000:af:00.0", 0, "0-3,8,10-11")
    assert pkg.numa_probe(str(tmp_path), "0000:AF:00.0") == (0, [0, 1, 2, 3, 8, 10, 11])


def test_numa_probe_says_unknown_instead_of_failing(pkg, tmp_path):
    make_sysfs(tmp_path, "0000:01:00.0", -1, "")                       # a one-node host or a VM
    assert pkg.numa_probe(str(tmp_path), "0000:01:00.0") == (-1, [])
    assert pkg.numa_probe(str(tmp_path), "0000:02:00.0") == (-1, [])   # no such device
    make_sysfs(tmp_path, "0000:03:00.0", 2, "3-1")                     # a malformed list binds nothing
    assert pkg.numa_probe(str(tmp_path), "0000:03:00.0") == (-1, [])
    # the caller's array may be shorter than the list: the count is still the whole list's
    make_sysfs(tmp_path, "0000:04:00.0", 3, "0-99")
    node, cpus = pkg.numa_probe(str(tmp_path), "0000:04:00.0", capacity=10)
    assert node == 3 and cpus == list(range(10))


def test_every_option_of_the_header_is_in_the_table_and_back(pkg):
    import re
    header = open(os.path.join(entry.ROOT, "include", "gdg.h")).read()
    block = header[header.index(" *   key "):header.index("int gdg_ctx_set_option")]
    documented = set(re.findall(r"^ \*   ([a-z_]+)\s", block, flags=re.M)) - {"key"}
    assert documented == set(pkg.option_names()), (sorted(documented ^ set(pkg.option_names())))


@pytest.mark.gpu
def test_options_round_trip_and_reject_what_they_should(pkg):
    ctx = pkg.Context(2, 1024)
    try:
        for key in pkg.option_names():
            v = ctx.get_option(key)
            ctx.set_option(key, v)                                     # its own value is always in range
            assert ctx.get_option(key) == v, key
        ctx.set_option("seg_wave_max_channels", 0)
        assert ctx.get_option("seg_wave_max_channels") == 0
        with pytest.raises(pkg.GdgError):
            ctx.set_option("no_such_option", 1)
        with pytest.raises(pkg.GdgError):
            ctx.set_option("fir_fused", 2)
        with pytest.raises(pkg.GdgError):
            ctx.set_option("copy_threads", 0)
    finally:
        ctx.close()

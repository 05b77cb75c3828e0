#!/usr/bin/env python3
"""Host-buffer entry points (gdg_process_staged / gdg_process) by number of PCIe channel groups (GDG_PCIE_GROUPS) and copy threads:
one child process per setting (the knobs are read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child():
    import bench
    import __graft_entry__ as entry
    pkg = entry.load_package()
    nch = int(os.environ.get("NCH", "512"))
    ctx = bench.make_context(pkg, nch, 8192, 0, 65536)
    r = bench.end_to_end(pkg, ctx, nch, 8192, 192000)
    print("   staged %.0f Msamples/s (%.3f ms, %.1f GB/s) | pageable %.0f Msamples/s (%.3f ms)" % (
        r["staged"]["value"], r["staged"]["ms_per_block"], r["staged"]["pcie_gbs_in_plus_out"], r["pageable"]["value"], r["pageable"]["ms_per_block"]), flush=True)
    ctx.close()

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
else:
    for env in ({"GDG_PCIE_GROUPS": "2"}, {"GDG_PCIE_WEIGHTS": "1,2,1"}, {"GDG_PCIE_WEIGHTS": "1,3"}, {"GDG_PCIE_WEIGHTS": "3,1"}, {"GDG_PCIE_WEIGHTS": "1,3,3,1"},
                {"GDG_PCIE_WEIGHTS": "1,2,2,2,1"}, {"GDG_PCIE_WEIGHTS": "1,6,1"}, {"GDG_PCIE_WEIGHTS": "2,3,3"}, {"GDG_PCIE_WEIGHTS": "1,1,2,2,1,1"},
                {"GDG_PCIE_WEIGHTS": "1,2,1", "NCH": "64"}, {"GDG_PCIE_WEIGHTS": "1,2,1", "NCH": "128"}, {"GDG_PCIE_GROUPS": "2", "NCH": "128"}):
        print(env, flush=True)
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)

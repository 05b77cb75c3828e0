#!/bin/bash
# bash profiles/run_pmc_window.sh <tag>: HBM traffic counters of the time-blocked loop (window_sweep.py 8, kernels alone), separate --pmc passes
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export GDG_DEVICE_GROUPS=1
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/profiles/window_sweep.py" 8 > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/profiles/window_sweep.py" 8 > /tmp/pw.log 2>&1
F=$(find /tmp/pmc_f -name '*.db' | head -1); W=$(find /tmp/pmc_w -name '*.db' | head -1)
{
  echo "# GDG_DEVICE_GROUPS=1 rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python profiles/window_sweep.py 8   (512 channels, 2 x 65536 taps, windows of 8 frames)"
  python - "$F" "$W" <<'PY'
import sqlite3, sys
for path, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection where counter_name=? group by kernel_name order by sum(value) desc", (counter,)).fetchall()
    print("\n%s per dispatch (KiB as reported; reads on gfx950: x 2, MI355X_MICROARCH.md)" % counter)
    for name, n, avg, mn, mx in rows[:8]:
        print("%-60s %6d %14.1f %14.1f %14.1f" % (name.split("(")[0][:60], n, avg, mn, mx))
PY
} > "$REPO/gpurun_out/${TAG}_window8_pmc.txt" 2>&1

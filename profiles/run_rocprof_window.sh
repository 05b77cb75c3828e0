#!/bin/bash
# bash profiles/run_rocprof_window.sh <tag>: kernel trace of the time-blocked device-resident loop (profiles/window_sweep.py, W = 16), alone and
# with two free-running channel groups
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for G in 1 2; do
  rm -rf /tmp/prof_w8
  GDG_DEVICE_GROUPS=$G rocprofv3 --kernel-trace --stats -d /tmp/prof_w8 -o w8 -- python "$REPO/profiles/window_sweep.py" 16 > /tmp/w8.log 2>&1
  W8=$(find /tmp/prof_w8 -name '*.db' | head -1)
  {
    echo "# GDG_DEVICE_GROUPS=$G (channel groups) rocprofv3 --kernel-trace --stats -- python profiles/window_sweep.py 16   (512 channels, 2 x 65536 taps, windows of 16 frames, 32 frames per channel, 3 passes)"
    grep "^W,\|^16," /tmp/w8.log
    python "$REPO/profiles/summarize_rocprof.py" "$W8"
  } > "$REPO/gpurun_out/${TAG}_window16_groups${G}_rocprof.txt" 2>&1
done

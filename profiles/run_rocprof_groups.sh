#!/bin/bash
# Run on the GPU box (gpurun):  bash profiles/run_rocprof_groups.sh <tag>
# kernel trace + stats of the DEFAULT bench command (timed region: two free-running channel groups; roofline pass: groups off), and
# (the time-blocked loop: profiles/run_rocprof_window.sh)
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity > /tmp/g.log 2>&1
G=$(find /tmp/prof_g -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras   (default: two free-running channel groups, opted into by bench.py)"
  echo "# bench line of the traced run:"; grep "^{" /tmp/g.log | tail -1
  python "$REPO/profiles/summarize_rocprof.py" "$G"
} > "$REPO/gpurun_out/${TAG}_bench_512ch_groups_rocprof.txt" 2>&1

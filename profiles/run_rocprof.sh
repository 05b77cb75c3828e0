#!/bin/bash
# Run on the GPU box (gpurun):  bash profiles/run_rocprof.sh <tag>
# kernel trace + stats of the default bench command, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as
# MI355X_MICROARCH.md prescribes; raw databases stay in /tmp, only the text summary goes to gpurun_out/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
# --channel-groups 1: every kernel runs alone (what the roofline pass of bench.py measures); the default command (two free-running channel
# groups in its timed region) is traced by profiles/run_rocprof_groups.sh
GROUPS_FLAG="--channel-groups ${GDG_BENCH_GROUPS:-1}"
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity $GROUPS_FLAG > /tmp/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-parity $GROUPS_FLAG > /tmp/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-parity $GROUPS_FLAG > /tmp/w.log 2>&1
KT=$(find /tmp/prof_kt -name '*.db' | head -1); F=$(find /tmp/prof_f -name '*.db' | head -1); W=$(find /tmp/prof_w -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity $GROUPS_FLAG   (+ --pmc FETCH_SIZE / WRITE_SIZE passes, --steps 5)"
  echo "# bench line of the traced run:"; grep "^{" /tmp/kt.log | tail -1
  python "$REPO/profiles/summarize_rocprof.py" "$KT" "$F" "$W"
} > "$REPO/gpurun_out/${TAG}_bench_512ch_rocprof.txt" 2>&1

#!/bin/bash
# Run on the GPU box (gpurun):  bash profiles/run_rocprof_tuner.sh <tag>
# kernel trace + separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of BASELINE config 5 (256 tuners + spatializer 256 -> 2)
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tprof_kt /tmp/tprof_f /tmp/tprof_w
rocprofv3 --kernel-trace --stats -d /tmp/tprof_kt -o kt -- python "$REPO/profiles/config5_rates.py" > /tmp/tkt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/tprof_f -o f -- python "$REPO/profiles/config5_rates.py" > /tmp/tf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/tprof_w -o w -- python "$REPO/profiles/config5_rates.py" > /tmp/tw.log 2>&1
KT=$(find /tmp/tprof_kt -name '*.db' | head -1); F=$(find /tmp/tprof_f -name '*.db' | head -1); W=$(find /tmp/tprof_w -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python profiles/config5_rates.py   (+ --pmc FETCH_SIZE / WRITE_SIZE passes); 256 channels per launch"
  tail -3 /tmp/tkt.log
  python "$REPO/profiles/summarize_rocprof.py" "$KT" "$F" "$W"
} > "$REPO/gpurun_out/${TAG}_config5_rocprof.txt" 2>&1

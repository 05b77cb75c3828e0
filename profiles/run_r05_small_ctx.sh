#!/bin/bash
# bash profiles/run_r05_small_ctx.sh <tag>: a GPU's share of the 512-channel job (64 / 128 channels) -- channel groups, hardware queues,
# and rocprofv3 kernel traces of the per-frame and the window path ("profiles/r05_64ch_rocprof.txt before/after")
set -u
TAG=${1:-r05a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
  for NCH in 64 128; do
    NCH=$NCH MODE=frame python $P
    NCH=$NCH MODE=frame KINDS=0 GPU_MAX_HW_QUEUES=8 python $P
    NCH=$NCH MODE=window NGROUPS_LIST=1,2,4 python $P
    NCH=$NCH MODE=window NGROUPS_LIST=1,2,4,8 KINDS=0 GPU_MAX_HW_QUEUES=8 python $P
  done
  NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1,2,4 python $P
} > "$OUT/${TAG}_small_ctx_groups.txt" 2>&1
for MODE in frame window; do
  rm -rf /tmp/prof_s
  NCH=64 MODE=$MODE NGROUPS_LIST=1 KINDS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name '*.db' | head -1)
  {
    echo "# NCH=64 MODE=$MODE NGROUPS_LIST=1 rocprofv3 --kernel-trace --stats -- python profiles/probes/small_ctx.py  (bench chain, 2 x 65536 taps, 192 kHz)"
    grep "groups:" /tmp/s.log
    python "$REPO/profiles/summarize_rocprof.py" "$DB"
  } > "$OUT/${TAG}_64ch_${MODE}_rocprof.txt" 2>&1
done
rm -rf /tmp/prof_s
NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
DB=$(find /tmp/prof_s -name '*.db' | head -1)
{ echo "# config 3 (64 ch, 96 kHz, 4x oversampling, 32768 taps), per-frame calls"; grep "groups:" /tmp/s.log; python "$REPO/profiles/summarize_rocprof.py" "$DB"; } > "$OUT/${TAG}_config3_rocprof.txt" 2>&1
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python "$REPO/profiles/probes/tuner_channels.py" > /tmp/s.log 2>&1
DB=$(find /tmp/prof_s -name '*.db' | head -1)
{ echo "# tuner analyses at 32 .. 256 channels"; cat /tmp/s.log | grep channels; python "$REPO/profiles/summarize_rocprof.py" "$DB"; } > "$OUT/${TAG}_tuner_rocprof.txt" 2>&1
echo done

#!/bin/bash
# bash profiles/run_r06_tile.sh <tag>: per-frame calls of few channels with a channel's frame on one / two workgroups (option seg_tile_max_channels)
set -u
TAG=${1:-r06p}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
  for NCH in 16 32 64 80 96; do
    for A in 0 1048576; do
      NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_tile_max_channels=$A" python $P
    done
  done
  for A in 0 1048576; do NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_tile_max_channels=$A" python $P; done
} > "$OUT/${TAG}_tile_ab.txt" 2>&1
for CH in bench config3; do
  rm -rf /tmp/prof_s
  NCH=64 MODE=frame CHAIN=$CH NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_tile_max_channels=1048576" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name '*.db' | head -1)
  { echo "# NCH=64 MODE=frame CHAIN=$CH seg_tile_max_channels=1048576"; grep "groups:" /tmp/s.log; python "$REPO/profiles/summarize_rocprof.py" "$DB"; } > "$OUT/${TAG}_64ch_${CH}_frame_rocprof.txt" 2>&1
done
echo done

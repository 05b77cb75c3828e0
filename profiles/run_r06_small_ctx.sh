#!/bin/bash
# bash profiles/run_r06_small_ctx.sh <tag>: per-frame calls of a GPU's share of the split job (32 .. 127 channels, BASELINE config 3) with round 6's
# three shapes off ("before": the reverbs' wet paths in their own unit, one workgroup per channel, the compressor in front of an oversampled
# shaper as a launch of its own) and at their defaults ("after"), and rocprofv3 kernel traces of both
set -u
TAG=${1:-r06a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
OFF="seg_reverb_ahead_max_channels=0,seg_tile_max_channels=0,seg_os_tiles_prefix=0"
{
  for NCH in 16 32 64 80 96 127; do
    NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS="$OFF" python $P
    NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  done
  NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 OPTIONS="$OFF" python $P
  NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 python $P
} > "$OUT/${TAG}_small_shards_ab.txt" 2>&1
for WHICH in before after; do
  O=$([ $WHICH = before ] && echo "$OFF" || echo "")
  for CH in bench config3; do
    rm -rf /tmp/prof_s
    NCH=64 MODE=frame CHAIN=$CH NGROUPS_LIST=1 KINDS=0 OPTIONS="$O" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
    DB=$(find /tmp/prof_s -name '*.db' | head -1)
    NAME=$([ $CH = bench ] && echo 64ch || echo config3)
    {
      echo "# NCH=64 MODE=frame CHAIN=$CH OPTIONS=$O rocprofv3 --kernel-trace --stats -- python profiles/probes/small_ctx.py"
      grep "groups:" /tmp/s.log
      python "$REPO/profiles/summarize_rocprof.py" "$DB"
    } > "$OUT/${TAG}_${NAME}_frame_rocprof_${WHICH}.txt" 2>&1
  done
done
echo done

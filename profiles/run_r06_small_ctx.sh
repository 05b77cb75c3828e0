#!/bin/bash
# bash profiles/run_r06_small_ctx.sh <tag>: per-frame calls of a GPU's share of the split job (64 / 96 / 127 channels, BASELINE config 3) with the
# reverbs' wet paths made ahead of the frame (option seg_reverb_ahead_max_channels) off and on, and rocprofv3 kernel traces of both
set -u
TAG=${1:-r06a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
  for NCH in 32 64 80 96 127; do
    for A in 0 200; do
      for H in 0 1; do
        NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_reverb_ahead_max_channels=$A,fir_premac_hosted=$H" python $P
      done
    done
  done
  for A in 0 200; do NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_reverb_ahead_max_channels=$A" python $P; done
} > "$OUT/${TAG}_reverb_ahead_ab.txt" 2>&1
for A in 0 200; do
  WHICH=$([ $A = 0 ] && echo before || echo after)
  rm -rf /tmp/prof_s
  NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_reverb_ahead_max_channels=$A,fir_premac_hosted=$([ $A = 0 ] && echo 0 || echo 1)" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name '*.db' | head -1)
  {
    echo "# NCH=64 MODE=frame OPTIONS=seg_reverb_ahead_max_channels=$A,fir_premac_hosted=$([ $A = 0 ] && echo 0 || echo 1) rocprofv3 --kernel-trace --stats -- python profiles/probes/small_ctx.py  (bench chain, 2 x 65536 taps, 192 kHz)"
    grep "groups:" /tmp/s.log
    python "$REPO/profiles/summarize_rocprof.py" "$DB"
  } > "$OUT/${TAG}_64ch_frame_rocprof_${WHICH}.txt" 2>&1
  rm -rf /tmp/prof_s
  NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 OPTIONS="seg_reverb_ahead_max_channels=$A" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name '*.db' | head -1)
  { echo "# config 3 (64 ch, 96 kHz, 4x oversampling, 32768 taps), per-frame calls, seg_reverb_ahead_max_channels=$A"; grep "groups:" /tmp/s.log; python "$REPO/profiles/summarize_rocprof.py" "$DB"; } > "$OUT/${TAG}_config3_frame_rocprof_${WHICH}.txt" 2>&1
done
echo done

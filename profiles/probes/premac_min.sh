set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
for NCH in 16 24 32 40 48; do
  for M in 384 1; do
    echo "# $NCH ch fir_premac_min_partitions=$M"
    OPTIONS=fir_premac_min_partitions=$M NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
    echo "# $NCH ch one amp fir_premac_min_partitions=$M"
    AMPS=1 OPTIONS=fir_premac_min_partitions=$M NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  done
done
for NCH in 64 72 80 88; do
  for A in 0 1048576; do
    echo "# $NCH ch seg_reverb_ahead_max_channels=$A"
    OPTIONS=seg_reverb_ahead_max_channels=$A NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  done
done
} > "$OUT/premac_min_ab.txt" 2>&1
echo done

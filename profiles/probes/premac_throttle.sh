set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
for NCH in 48 64 80 96 112 128; do
  for U in 7 4 8; do
    for LDS in 0 16384 32768 49152 65536; do
      echo "# $NCH ch premac unroll $U lds $LDS"
      GDG_EXP_PREMAC_UNROLL=$U GDG_EXP_PREMAC_LDS=$LDS NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
    done
  done
done
for NCH in 144 160 192 256; do
  echo "# $NCH ch fused unroll 0 lds 0"
  NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  for LDS in 0 32768 49152 65536; do
      echo "# $NCH ch split unroll 7 lds $LDS"
      GDG_EXP_PREMAC_LDS=$LDS OPTIONS=fir_split_max_channels=1048576 NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  done
done
for LDS in 0 32768; do
      echo "# config3 64 ch premac unroll 7 lds $LDS"
      GDG_EXP_PREMAC_LDS=$LDS NCH=64 CHAIN=config3 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
done
} > "$OUT/premac_throttle_ab2.txt" 2>&1
echo done

set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
for R in 1 2; do
for T in old new; do
  P=$([ $T = old ] && echo "$REPO/.ab_old/profiles/probes/small_ctx.py" || echo "$REPO/profiles/probes/small_ctx.py")
  for NCH in 64 96 128; do
    for F in 0 1; do
      echo "# $T: bench $NCH ch handoff=$F"; OPTIONS=fir_premac_in_memory_handoff=$F NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
    done
  done
  echo "# $T: config3 64 ch"; CHAIN=config3 NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# $T: bench 512 ch"; NCH=512 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
done
done
} > "$OUT/setprio_ab.txt" 2>&1
echo done

set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
for R in 1 2 3; do
for T in old new; do
  P=$([ $T = old ] && echo "$REPO/.ab_old/profiles/probes/small_ctx.py" || echo "$REPO/profiles/probes/small_ctx.py")
  echo "# $T: config3 64 ch"; CHAIN=config3 NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# $T: config3 112 ch"; CHAIN=config3 NCH=112 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# $T: bench 32 ch (no premac: mac of 8 terms on the main stream)"; NCH=32 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# $T: bench 16 ch"; NCH=16 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
done
done
} > "$OUT/mac_tail_ab.txt" 2>&1
echo done

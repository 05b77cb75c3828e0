set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
for NCH in 64 96 128; do
  echo "# $NCH channels: no masks"
  NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  for K in 32 64 96 128; do
    echo "# side stream on $K CUs, main stream everywhere"
    GDG_EXP_CU_SIDE=$K NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
    echo "# side stream on $K CUs, main stream on the other $((256-K))"
    GDG_EXP_CU_SIDE=$K GDG_EXP_CU_MAIN=$K NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  done
done
} > "$OUT/cu_split_ab.txt" 2>&1
rm -rf /tmp/prof_s
GDG_EXP_CU_SIDE=64 GDG_EXP_CU_MAIN=64 NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
DB=$(find /tmp/prof_s -name '*.db' | head -1)
{ grep "groups:" /tmp/s.log; python "$REPO/profiles/summarize_rocprof.py" "$DB"; } > "$OUT/cu_split_64ch_rocprof.txt" 2>&1
echo done

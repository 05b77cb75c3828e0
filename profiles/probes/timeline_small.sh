# one step's kernel timeline of a small shard's per-frame calls: NCH=64 bash profiles/probes/timeline_small.sh  (gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for NCH in ${NCHS:-64 128}; do
rm -rf /tmp/prof_t
NCH=$NCH MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS=${OPTIONS:-} rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python "$REPO/profiles/probes/small_ctx.py" > /tmp/t.log 2>&1
CSV=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
{ echo "# NCH=$NCH OPTIONS=${OPTIONS:-}"; grep "groups:" /tmp/t.log; python "$REPO/profiles/probes/timeline.py" "$CSV" 24; } > "$OUT/timeline_${NCH}ch.txt" 2>&1
done
echo done

set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
: > "$OUT/premac_lds_prof.txt"
for CFG in "128 65536" "128 32768" "48 65536"; do
set -- $CFG
for LDS in 0 16384 49152; do
rm -rf /tmp/prof_s
TAPS=$2 OPTIONS=fir_premac_lds_bytes=$LDS NCH=$1 MODE=frame NGROUPS_LIST=1 KINDS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
DB=$(find /tmp/prof_s -name '*.db' | head -1)
{ echo "# NCH=$1 TAPS=$2 LDS=$LDS"; grep "groups:" /tmp/s.log; python "$REPO/profiles/summarize_rocprof.py" "$DB" | head -8; } >> "$OUT/premac_lds_prof.txt" 2>&1
done
done
echo done

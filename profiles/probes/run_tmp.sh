cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_b
NGROUPS=2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $REPO/profiles/probes/batch_kinds.py > /tmp/b.log 2>&1
DB=$(find /tmp/prof_b -name '*.db' | head -1)
{
  echo "# NGROUPS=2 rocprofv3 --kernel-trace --stats -- python profiles/probes/batch_kinds.py   (gdg_batch_run: 512 x 16-bit files of 128 blocks -> 515 x 24-bit files, W = 16, two channel groups; four runs in the trace)"
  grep "run:" /tmp/b.log
  python $REPO/profiles/summarize_rocprof.py "$DB"
} > $REPO/gpurun_out/r05_batch_run_rocprof.txt 2>&1

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_os_tiles.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05l_tests.txt
for MODE in frame window; do
rm -rf /tmp/prof_s
NCH=64 MODE=$MODE CHAIN=config3 NGROUPS_LIST=1 KINDS=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python profiles/probes/small_ctx.py > /tmp/s.log 2>&1
DB=$(find /tmp/prof_s -name '*.db' | head -1)
{ echo "# config 3 (64 ch, 96 kHz, 4x oversampling, 32768 taps), MODE=$MODE, after: the oversampled overdrive as a launch of its own (os_tiles_kernel)"; grep "groups:" /tmp/s.log; python profiles/summarize_rocprof.py "$DB"; } > gpurun_out/r05_config3_${MODE}_rocprof_after.txt 2>&1
done

cd $GRAFT_REPO_ROOT
GDG_BATCH_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r05z_bench_trace.json 2> gpurun_out/r05z_bench_trace.err
grep "^\[batch\]" gpurun_out/r05z_bench_trace.err | tail -40 > gpurun_out/r05z_bench_trace.txt
timeout 300 python profiles/probes/batch_kinds.py > gpurun_out/r05z_batch_kinds.txt 2>&1

cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r05_final3.json 2> gpurun_out/bench_r05_final3.err

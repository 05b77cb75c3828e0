cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05z_pytest_gpu.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_r05_final2.json 2> gpurun_out/bench_r05_final2.err

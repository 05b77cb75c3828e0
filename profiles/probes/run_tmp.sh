cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_end_to_end.py -x -q > gpurun_out/r05x_tests.txt 2>&1

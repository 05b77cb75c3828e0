cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_end_to_end.py tests/test_host_mirror.py -x -q -m gpu > gpurun_out/r05x_tests.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "batch" >> gpurun_out/r05x_tests.txt 2>&1
GDG_BATCH_TRACE=1 timeout 300 python profiles/probes/batch_kinds.py > gpurun_out/r05x_batch_trace.txt 2>&1

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_end_to_end.py tests/test_host_mirror.py -x -q -m gpu > gpurun_out/r05_final_e2e.txt 2>&1
NUMA_MODES="2 0" timeout 900 bash profiles/run_numa.sh r05b

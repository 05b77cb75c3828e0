cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05p_pytest_gpu.txt 2>&1
for n in 96 128 160; do NCH=$n MODE=frame NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py; done > gpurun_out/r05p_small.txt 2>&1

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r05c_pytest_gpu.txt
for NCH in 64; do NCH=$NCH MODE=window NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py; done > gpurun_out/r05j_light.txt 2>&1

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05c_pytest_gpu.txt
P=profiles/probes/small_ctx.py
{
NCH=64 MODE=window CHAIN=config3 NGROUPS_LIST=1 KINDS=0 timeout 300 python $P
NCH=64 MODE=frame CHAIN=config3 NGROUPS_LIST=1 KINDS=0 timeout 300 python $P
} > gpurun_out/r05f_config3.txt 2>&1

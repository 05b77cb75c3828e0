cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_premac.py tests/test_gpu_overlap.py tests/test_gpu_boundary.py tests/test_options_numa.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05e_tests.txt
P=profiles/probes/small_ctx.py
{
NCH=64 MODE=frame NGROUPS_LIST=1,2,1 KINDS=0 timeout 300 python $P
} > gpurun_out/r05e_premac.txt 2>&1

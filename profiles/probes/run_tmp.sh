cd $GRAFT_REPO_ROOT
P=profiles/probes/small_ctx.py
{
NCH=128 MODE=frame NGROUPS_LIST=1,1 timeout 300 python $P
NCH=256 MODE=frame NGROUPS_LIST=1 timeout 300 python $P
} > gpurun_out/r05i_128.txt 2>&1

cd $GRAFT_REPO_ROOT
P=profiles/probes/small_ctx.py
{
NCH=512 MODE=window NGROUPS_LIST=1,2,2 KINDS=0 timeout 600 python $P
GDG_SEG_WAVE_MAX=512 NCH=512 MODE=window NGROUPS_LIST=1,2,2 KINDS=0 timeout 600 python $P
} > gpurun_out/r05n_512_wave.txt 2>&1

cd $GRAFT_REPO_ROOT
for n in 32 64 96; do GDG_SEG_FAST_MIN=1 NCH=$n MODE=window NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py; done > gpurun_out/r05v_small2.txt 2>&1
for n in 96; do NCH=$n MODE=window NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py; done >> gpurun_out/r05v_small2.txt 2>&1

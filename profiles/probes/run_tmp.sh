cd $GRAFT_REPO_ROOT
for g in 1 2; do echo "NGROUPS=$g"; NGROUPS=$g timeout 300 python profiles/probes/batch_kinds.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05z_batch_groups.txt

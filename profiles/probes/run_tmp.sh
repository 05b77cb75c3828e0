cd $GRAFT_REPO_ROOT
timeout 300 python profiles/config5_rates.py 2>&1 | grep -v amdgpu > gpurun_out/r05_spat_nolds.txt
timeout 300 python profiles/config5_rates.py 2>&1 | grep -v amdgpu >> gpurun_out/r05_spat_nolds.txt

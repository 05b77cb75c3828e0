cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_premac.py tests/test_gpu_boundary.py -x -q > gpurun_out/r05_chain_tests.txt 2>&1
for n in 48 64 96 128; do
  for sc in 1 0; do echo -n "GDG_FIR_SMALL_CHAIN=$sc "; GDG_FIR_SMALL_CHAIN=$sc NCH=$n MODE=frame NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py 2>&1 | grep -v amdgpu; done
done > gpurun_out/r05_chain_small.txt 2>&1

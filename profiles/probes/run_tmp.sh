cd $GRAFT_REPO_ROOT
SECONDS=100 timeout 400 python profiles/probes/wave_soak.py > gpurun_out/r05_soak_long.txt 2>&1
echo "exit $?" >> gpurun_out/r05_soak_long.txt

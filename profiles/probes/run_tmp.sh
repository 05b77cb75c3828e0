cd $GRAFT_REPO_ROOT
timeout 900 python profiles/probes/fuzz_soak.py 6000 700 > gpurun_out/r05t_fuzz_soak2.txt 2>&1
echo "exit $?" >> gpurun_out/r05t_fuzz_soak2.txt

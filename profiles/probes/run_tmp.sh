cd $GRAFT_REPO_ROOT
s=$(date +%s.%N)
timeout 900 python bench.py > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_r05_final.err
e=$(date +%s.%N)
echo "bench.py wall: $(echo "$e - $s" | bc) s, exit $?" > gpurun_out/bench_r05_final.time

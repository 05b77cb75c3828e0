cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" > gpurun_out/r05_final_smoke.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05_final_pytest_gpu.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_r05_final4.json 2> gpurun_out/bench_r05_final4.err

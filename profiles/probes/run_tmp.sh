cd $GRAFT_REPO_ROOT
timeout 300 python profiles/probes/spat_var.py > gpurun_out/r05u_spat_var.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_tuner_spatializer.py -x -q -k spatializer > gpurun_out/r05u_spat_tests.txt 2>&1
for v in 2 4 5 6; do GDG_SPAT_VAR=$v timeout 300 python -m pytest tests/test_gpu_tuner_spatializer.py tests/test_gpu_advice_r03.py -x -q -k "spatializer" 2>&1 | tail -1 >> gpurun_out/r05u_spat_tests.txt; done

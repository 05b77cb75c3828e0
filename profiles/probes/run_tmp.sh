cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_tuner_spatializer.py -x -q > gpurun_out/r05s_tuner_tests.txt 2>&1
timeout 600 python profiles/probes/tuner_pairs.py > gpurun_out/r05s_tuner_pairs.txt 2>&1

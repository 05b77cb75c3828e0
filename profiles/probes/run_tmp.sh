cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_tuner_spatializer.py tests/test_gpu_fuzz.py -x -q -k "tuner" > gpurun_out/r05_tuner_tests.txt 2>&1
timeout 300 python profiles/probes/tuner_channels.py > gpurun_out/r05_tuner_channels.txt 2>&1

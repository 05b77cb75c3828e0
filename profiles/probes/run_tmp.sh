cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tuner_spatializer.py tests/test_gpu_fuzz.py tests/test_host_mirror.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r05k_tests.txt
python profiles/probes/tuner_channels.py > gpurun_out/r05k_tuner.txt 2>&1

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_segf.py tests/test_gpu_fuzz.py tests/test_gpu_end_to_end.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05b_tests.txt
P=profiles/probes/small_ctx.py
{
for NCH in 32 64 96 128; do
  NCH=$NCH MODE=window NGROUPS_LIST=1 timeout 300 python $P
done
GDG_SEG_WAVE_MAX=0 NCH=32 MODE=window NGROUPS_LIST=1 timeout 300 python $P
GDG_SEG_WAVE_MAX=0 NCH=96 MODE=window NGROUPS_LIST=1 timeout 300 python $P
} > gpurun_out/r05b_wave.txt 2>&1

#!/bin/bash
# A/B of the one-buffer 8192-point transforms (GDG_FFT_HALF_LDS bits) on one box: window tests for the bits, then the W = 16 sweep per setting
cd "$(dirname "$0")/../.."
SETTINGS="${SETTINGS:-0 2 10}"
for h in $SETTINGS; do
  echo "== GDG_FFT_HALF_LDS=$h: window + chain tests"
  GDG_FFT_HALF_LDS=$h python -m pytest tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -2
done
for rep in 1 2 3; do
  for h in $SETTINGS; do
    echo "== GDG_FFT_HALF_LDS=$h rep $rep"
    GDG_FFT_HALF_LDS=$h python profiles/window_sweep.py 16 | tail -1
  done
done

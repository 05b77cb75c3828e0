#!/bin/bash
# same-box A/B of two builds of libgdg.so: $1 = the other library (copied over lib/libgdg.so for its turn); prints the segment breakdown (two-per-CU
# and general kernel), the W = 16 loop and the small-shard legs for both
cd "$(dirname "$0")/../.."
OLD=$1
cp go-dsp-guitar_amd/lib/libgdg.so /tmp/libgdg_new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp "$OLD" go-dsp-guitar_amd/lib/libgdg.so; else cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so; fi
    echo "== $which rep $rep: two-per-CU kernel"
    python profiles/seg_breakdown.py 2>&1 | grep -E "copy|compressor|overdrive |tone_stack|chorus|cabinet|reverb|seg0|seg1"
    python profiles/probes/window_groups.py 2>&1 | head -2
    echo "== $which rep $rep: general kernel (GDG_SEG_FAST=0)"
    GDG_SEG_FAST=0 python profiles/seg_breakdown.py 2>&1 | grep -E "copy|compressor|overdrive|tone_stack|chorus|cabinet|reverb|flanger|delay|seg0|seg1"
    python profiles/probes/fast_min_ab.py 2>&1 | tail -3
  done
done
cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so

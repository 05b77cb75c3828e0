cd $GRAFT_REPO_ROOT
cp go-dsp-guitar_amd/lib/libgdg.so /tmp/libgdg_new.so
{
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp profiles/probes/bin/libgdg_old.so go-dsp-guitar_amd/lib/libgdg.so; else cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so; fi
    echo "== $which rep $rep"
    python profiles/seg_breakdown.py reverb "seg1 of bench" "seg0 of bench" 2>&1 | grep -E "reverb|seg0|seg1"
    python profiles/seg_breakdown.py --cold reverb "seg1 of bench" 2>&1 | grep -E "reverb|seg1"
  done
done
} > gpurun_out/r05_reverb_tail_ab.txt 2>&1
cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so

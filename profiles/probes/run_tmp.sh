cd $GRAFT_REPO_ROOT
cp go-dsp-guitar_amd/lib/libgdg.so /tmp/libgdg_new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp profiles/probes/bin/libgdg_old.so go-dsp-guitar_amd/lib/libgdg.so; else cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so; fi
    echo "== $which rep $rep"
    python profiles/seg_breakdown.py 2>&1 | grep -E "copy|compressor|overdrive |tone_stack|chorus|cabinet|reverb|seg0|seg1"
    timeout 300 python bench.py --steps 20 --warmup 25 --no-cpu-baseline --no-extras --no-parity --channel-groups 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['roofline']['kernels_ms']
        print('bench 1 group: ms_per_step %.4f segment avg %.2f us' % (d['ms_per_step'], 1e3 * k['segment']['avg_ms']))
"
    NCH=64 MODE=window NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py 2>&1 | grep -v amdgpu
    NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 timeout 300 python profiles/probes/small_ctx.py 2>&1 | grep -v amdgpu
  done
done > gpurun_out/r05_prefetch_ab.txt 2>&1
cp /tmp/libgdg_new.so go-dsp-guitar_amd/lib/libgdg.so

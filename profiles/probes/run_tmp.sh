cd $GRAFT_REPO_ROOT
{
echo "== WAVE (default)"; NCH=64 python profiles/seg_breakdown.py --window=16 "(copy)" compressor overdrive tone_stack chorus cabinet reverb "seg0 of bench" "seg1 of bench" 2>&1 | grep -v "^chain"
echo "== walk (GDG_SEG_WAVE_MAX=0)"; GDG_SEG_WAVE_MAX=0 NCH=64 python profiles/seg_breakdown.py --window=16 "(copy)" compressor overdrive tone_stack chorus cabinet reverb "seg0 of bench" "seg1 of bench" 2>&1 | grep -v "^chain"
} > gpurun_out/r05h_seg_window_64.txt 2>&1

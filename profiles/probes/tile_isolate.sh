cd /tmp; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/profiles/probes/small_ctx.py
for O in "seg_tile_max_channels=0,seg_reverb_ahead_max_channels=0" "seg_tile_max_channels=80,seg_reverb_ahead_max_channels=0" "seg_tile_max_channels=0,seg_reverb_ahead_max_channels=0,fir_premac=0" "seg_tile_max_channels=80,seg_reverb_ahead_max_channels=0,fir_premac=0" "seg_tile_max_channels=80,fir_premac=0" "seg_tile_max_channels=0,fir_premac=0"; do
  rm -rf /tmp/prof_s
  NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 OPTIONS="$O" rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $P > /tmp/s.log 2>&1
  DB=$(find /tmp/prof_s -name '*.db' | head -1)
  echo "# $O"; grep "groups:" /tmp/s.log; python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py "$DB" | grep "seg\|kernel  "
done

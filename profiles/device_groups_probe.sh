for g in 1 2 4; do echo "GDG_DEVICE_GROUPS=$g"; GDG_DEVICE_GROUPS=$g python profiles/window_sweep.py 2>&1 | tail -4 | cut -d, -f1-4; done
for g in 1 2 4; do echo "GDG_DEVICE_GROUPS=$g"; GDG_DEVICE_GROUPS=$g python profiles/channels_sweep.py 2>/dev/null | cut -d, -f1-4; done

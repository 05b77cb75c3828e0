#!/bin/bash
# bash profiles/run_numa.sh <tag>: the host-buffer paths with the caller bound to either socket, option "numa" on and off
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out/host_path_numa_${TAG}.txt"
mkdir -p "$REPO/gpurun_out"
cd "$REPO"
{
  echo "# $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | xargs), $(nproc) logical CPUs; nodes: $(cat /sys/devices/system/node/online 2>/dev/null)"
  for n in /sys/devices/system/node/node*; do echo "# $(basename $n): cpus $(cat $n/cpulist)"; done
  python - <<'PY'
import __graft_entry__ as e, ctypes as C
hip = C.CDLL("libamdhip64.so"); b = C.create_string_buffer(64); hip.hipDeviceGetPCIBusId(b, 64, 0)
print("# device 0 = %s: numa_probe -> %s" % (b.value.decode(), e.load_package().numa_probe("/sys", b.value.decode())[0]))
PY
  N0=$(cat /sys/devices/system/node/node0/cpulist 2>/dev/null)
  N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
  for CPUS in "$N0" "$N1"; do
    [ -z "$CPUS" ] && continue
    for NUMA in ${NUMA_MODES:-1 0}; do
      echo "== caller bound to cpus $CPUS, option numa = $NUMA (env GDG_NUMA: the debug override of the option's default)"
      GDG_NUMA=$NUMA taskset -c "$CPUS" python profiles/host_path_rate.py 2>&1 | grep -v "^entry"
      GDG_NUMA=$NUMA taskset -c "$CPUS" python profiles/probes/batch_kinds.py 2>&1 | grep "plain run"
    done
  done
} > "$OUT" 2>&1

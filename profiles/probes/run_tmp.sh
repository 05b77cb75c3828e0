cd $GRAFT_REPO_ROOT
NUMA_MODES="2" timeout 1500 bash profiles/run_numa.sh r05_mode2
{
N1=$(cat /sys/devices/system/node/node1/cpulist)
echo "== caller on node1, numa 1, 16 copy threads"
GDG_COPY_THREADS=16 GDG_NUMA=1 taskset -c "$N1" python profiles/probes/batch_kinds.py 2>&1 | grep "plain run"
echo "== caller on node1, numa 0, 16 copy threads"
GDG_COPY_THREADS=16 GDG_NUMA=0 taskset -c "$N1" python profiles/probes/batch_kinds.py 2>&1 | grep "plain run"
echo "== unbound caller, numa 1 / 0 / 2"
for M in 1 0 2; do GDG_NUMA=$M python profiles/probes/batch_kinds.py 2>&1 | grep "plain run"; GDG_NUMA=$M python profiles/host_path_rate.py 2>&1 | grep -v "^entry"; done
} >> gpurun_out/host_path_numa_r05_mode2.txt 2>&1

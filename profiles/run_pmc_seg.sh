#!/bin/bash
# Run on the GPU box:  bash profiles/run_pmc_seg.sh <tag>     VALU issue counters of the segment kernel (bench workload, channel groups off)
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-parity --channel-groups 1"
OUT="$REPO/gpurun_out/${TAG}_seg_valu_pmc.txt"
echo "# rocprofv3 --pmc <set> -- $CMD   (one pass per counter set)" > "$OUT"
i=0
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $SET -d /tmp/pmc_$i -o p -- $CMD > /tmp/pmc_$i.log 2>&1 || { echo "# set '$SET' failed:" >> "$OUT"; tail -3 /tmp/pmc_$i.log >> "$OUT"; continue; }
  DB=$(find /tmp/pmc_$i -name '*.db' | head -1)
  echo "# set: $SET" >> "$OUT"
  python "$REPO/profiles/pmc_summary.py" "$DB" seg_kernel >> "$OUT" 2>&1
  python "$REPO/profiles/pmc_summary.py" "$DB" segf_kernel >> "$OUT" 2>&1
  python "$REPO/profiles/pmc_summary.py" "$DB" fir_inv_kernel >> "$OUT" 2>&1
done

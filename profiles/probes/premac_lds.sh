set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="$REPO/profiles/probes/small_ctx.py"
{
for R in 1 2; do
for LDS in -1 0 16384 49152; do
  echo "# long filter 1048576 taps, 64 ch, one amp, 4 distinct IRs: fir_premac_lds_bytes=$LDS"
  TAPS=1048576 AMPS=1 DISTINCT=4 OPTIONS=fir_premac_lds_bytes=$LDS NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# 262144 taps, 64 ch, two amps: fir_premac_lds_bytes=$LDS"
  TAPS=262144 OPTIONS=fir_premac_lds_bytes=$LDS NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# 32768 taps, 128 ch, two amps: fir_premac_lds_bytes=$LDS"
  TAPS=32768 OPTIONS=fir_premac_lds_bytes=$LDS NCH=128 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
  echo "# config3: fir_premac_lds_bytes=$LDS"
  CHAIN=config3 OPTIONS=fir_premac_lds_bytes=$LDS NCH=64 MODE=frame NGROUPS_LIST=1 KINDS=0 python $P
done
done
} > "$OUT/premac_lds_ab.txt" 2>&1
echo done

#!/bin/bash
# Kernel-trace + PMC passes (separately, as MI355X_MICROARCH.md prescribes) over tools/profile_paths.py, one workload at a time.
#   [REPS=220] tools/profile_paths.sh <out_dir> <workload> [<workload> ...]      -> <out_dir>/<workload>/{kernel_stats.csv, pmc_summary.json}
# REPS: launches of the kernel-trace pass (default 20; the counter passes always take 6)
set -o pipefail
OUT=${1:?out dir}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; export TMPDIR=/tmp
for w in "$@"; do
  D="$OUT/$w"; mkdir -p "$D/raw"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/raw/kt" -o p -- python3 tools/profile_paths.py --what $w --reps ${REPS:-20} > "$D/raw/kt.log" 2>&1 || { echo "$w: trace failed"; tail -5 "$D/raw/kt.log"; continue; }
  cp "$(find "$D/raw/kt" -name 'p_kernel_stats.csv' | head -1)" "$D/kernel_stats.csv"
  i=0
  for counters in "FETCH_SIZE" "WRITE_SIZE" \
      "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    i=$((i + 1))
    rocprofv3 --pmc $counters --output-format csv -d "$D/raw/pmc$i" -o p -- python3 tools/profile_paths.py --what $w --reps 6 > "$D/raw/pmc$i.log" 2>&1 || { echo "$w: pmc pass $i failed"; tail -3 "$D/raw/pmc$i.log"; }
  done
  python3 tools/pmc_summary.py "$D/raw" "$D/pmc_summary.json" > "$D/summary.txt" 2>&1
  rm -rf "$D/raw"
  echo "== $w"; head -12 "$D/kernel_stats.csv" | cut -c1-200
done

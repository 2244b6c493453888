#!/bin/bash
# Profile bench.py on the GPU box: one kernel-trace pass (per-kernel durations) and separate PMC passes, as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass; counters never together with tracing
# of other domains).  Usage (on the GPU box, from the repo root):
#   tools/profile_gpu.sh <out_dir> [bench.py arguments ...]
# Writes <out_dir>/{kernel_stats.csv, pmc_summary.json, bench_line.json}; raw rocprofv3 output stays in <out_dir>/raw.
set -o pipefail
OUT=${1:?out dir}; shift
BENCH_ARGS=${@:---steps 50 --warmup 5 --cpu-scans 0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT/raw"
export TMPDIR=/tmp
cd "$ROOT"
# PASSES (default: all): which parts to run in this call -- "kt" (bench line + kernel trace) and / or PMC pass numbers 1..5; a GPU call
# has a time limit, and seven runs of bench.py with its side measurements do not fit one
PASSES=${PASSES:-kt 1 2 3 4 5}
if [[ " $PASSES " == *" kt "* ]]; then
python3 bench.py $BENCH_ARGS > "$OUT/bench_line.json" 2> "$OUT/raw/bench.err" || exit 1
echo "bench line done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw/kt" -o p -- python3 bench.py $BENCH_ARGS > "$OUT/raw/kt.log" 2>&1 || exit 1
cp "$(find "$OUT/raw/kt" -name 'p_kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
echo "kernel trace done"
fi
PMC_ARGS="--steps 6 --warmup 2 --cpu-scans 0 ${PMC_EXTRA:-}"
i=0
for counters in \
    "FETCH_SIZE" \
    "WRITE_SIZE" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
    "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  [[ " $PASSES " == *" $i "* ]] || continue
  rocprofv3 --pmc $counters --output-format csv -d "$OUT/raw/pmc$i" -o p -- python3 bench.py $PMC_ARGS > "$OUT/raw/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$OUT/raw/pmc$i.log"; }
  echo "pmc pass $i done"
done
python3 tools/pmc_summary.py "$OUT/raw" "$OUT/pmc_summary.json"
# (the raw traces are tens of megabytes: only the summaries travel back)
rm -rf "$OUT"/raw/pmc[0-9] "$OUT/raw/kt"

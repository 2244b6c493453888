#!/bin/bash
# Quick PMC comparison of forms of the step kernel: for each --step-var value, the SQ / LDS passes of tools/profile_gpu.sh
# on a short bench.py run (no tracing, counters only).   tools/pmc_quick.sh <out_dir> <var> [<var> ...]
set -o pipefail
OUT=${1:?out dir}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; export TMPDIR=/tmp
for v in "$@"; do
  D="$OUT/var$v"; mkdir -p "$D/raw"
  i=0
  for counters in \
      "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
    i=$((i + 1))
    rocprofv3 --pmc $counters --output-format csv -d "$D/raw/pmc$i" -o p -- python3 bench.py --steps 6 --warmup 2 --cpu-scans 0 --no-extras --step-var $v > "$D/raw/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$D/raw/pmc$i.log"; }
  done
  python3 tools/pmc_summary.py "$D/raw" "$D/pmc_summary.json" > "$D/summary.txt" 2>&1
  python3 - "$D/pmc_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'consistency_step' in k:
        print(k[:90]); print({a: round(b, 1) for a, b in v['counters'].items()})
PY
done

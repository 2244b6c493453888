#!/bin/bash
# quick SQ counters for a profile_paths workload: pmcq.sh <out> <what> [extra args]
OUT=$1; W=$2; shift 2
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
mkdir -p $OUT/raw
i=0
for counters in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $OUT/raw/pmc$i -o p -- python3 tools/profile_paths.py --what $W --reps 6 "$@" > $OUT/raw/pmc$i.log 2>&1 || tail -3 $OUT/raw/pmc$i.log
done
python3 tools/pmc_summary.py $OUT/raw $OUT/pmc_summary.json > /dev/null 2>&1
python3 - $OUT/pmc_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'consistency_step' in k:
        print(k[:80]); print({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != 'counters'}); print({a: round(b, 1) for a, b in v['counters'].items()})
PY

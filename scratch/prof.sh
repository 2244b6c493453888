set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/stats -o s -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-scans 0 > $R/gpurun_out/r01c_stats.log 2>&1
echo stats done
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $R/gpurun_out/r01c/p$i -o p -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-scans 0 > $R/gpurun_out/r01c_p$i.log 2>&1
  echo "pass $i done"
done <<'LST'
FETCH_SIZE
WRITE_SIZE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum
TA_TA_BUSY_sum
LST

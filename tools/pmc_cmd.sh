#!/bin/bash
# tools/pmc_cmd.sh <tag> <kernel-name-pattern> <python script + args ...>
# rocprofv3 counter passes focused on the memory path (TA / TCP / TCC / translation), one group per pass, counters in
# their own runs with --kernel-trace only, at most 4 counters of one hardware block per pass (TCC and TA have 4 slots).
# GPU box only.  Summary -> gpurun_out/<tag>/summary.txt.  A failed pass fails the script (after the summary is written).
set -u
FAILED=0
TAG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
  "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  "TCC_TAG_STALL_sum" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_WRITEBACK_sum" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i FAILED: $grp"; FAILED=1; }
done
python3 $R/tools/pmc_summary.py $OUT $PAT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
if [ $FAILED -ne 0 ]; then echo "pmc_cmd.sh: at least one counter pass failed (see p*.log)"; exit 1; fi

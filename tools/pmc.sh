#!/bin/bash
# tools/pmc.sh <tag> [kernel-name-pattern] -- rocprofv3 counter passes over a short bench.py run (GPU box only).
# PMC_ARGS overrides the bench.py arguments (e.g. "--dist-selftest --virtual-world 8 ..." for the rebuild kernel).
# Counters are collected in their own runs (no tracing domains besides --kernel-trace), one group per pass.
set -u
FAILED=0
TAG=${1:-pmc}
PAT=${2:-trace_kernel}
# PMC_SCENE=<scene> profiles bench.py --scene <scene>; the summary (pmc.json) is then meant for profiles/pmc_<scene>.json
SCENE_ARGS=${PMC_SCENE:+--scene $PMC_SCENE}
# a caller's PMC_ARGS keeps the scene too (it used to drop it silently: a summary filed under the wrong scene name)
# --serial: one un-overlapped trace launch per step (inside the scan pipeline the launches carry the previous steps' scatter
# workgroups in front: not what the roofline prices); short timed phase, the counters are per launch
ARGS="${PMC_ARGS:---serial --steps 3 --warmup 1 --min-seconds 0.05 --no-cpu-baseline --no-caller-path} $SCENE_ARGS"
export PMC_ARGS="$ARGS"
# PMC_SCRIPT="tools/rebuild_time.py 8": profile another script of the repo instead of bench.py (its own arguments included)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
TARGET="$R/bench.py $ARGS"
[ -n "${PMC_SCRIPT:-}" ] && TARGET="$R/$PMC_SCRIPT"
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $TARGET > $OUT/p$i.log 2>&1 || { echo "pass $i FAILED: $grp"; FAILED=1; }
done
python3 $R/tools/pmc_summary.py $OUT $PAT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
if [ $FAILED -ne 0 ]; then echo "pmc.sh: at least one counter pass failed (see p*.log)"; exit 1; fi

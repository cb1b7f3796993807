#!/bin/bash
# tools/pmc_occ.sh <tag> -- what limits wave launch / residency of the trace kernel (GPU box only)
set -u
TAG=${1:-pmc_occ}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in \
  "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN" \
  "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN" \
  "SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
  "SPI_CSN_BUSY SPI_CSN_WINDOW_VALID SPI_CSN_WAVE SPI_CSN_NUM_THREADGROUPS" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

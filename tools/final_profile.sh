#!/bin/bash
# tools/final_profile.sh <tag> -- the evidence set DESIGN.md quotes, from ONE box: bench line, rocprofv3 kernel
# statistics of the same command, PMC passes (tools/pmc.sh), the memory-path PMC of the trace kernel (tools/pmc_cmd.sh
# on bench.py) and the single-GPU rehearsal of the N-GPU step.  GPU box only; outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1 || echo "stats run failed"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo "stats done"
cd $R
tools/pmc.sh $TAG/pmc > /dev/null 2>&1
cp gpurun_out/$TAG/pmc/summary.txt $OUT/pmc_summary.txt 2>/dev/null
cp gpurun_out/$TAG/pmc/pmc.json $OUT/pmc.json 2>/dev/null
echo "pmc done"
VW_LIST="1 2 4 8" tools/vw_sweep.sh > $OUT/vw_sweep.txt 2>&1
cat $OUT/vw_sweep.txt
head -4 $OUT/kernel_stats.csv | cut -c1-160
python3 -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['cpu_baseline']['value'])"

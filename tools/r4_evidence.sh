#!/bin/bash
# tools/r4_evidence.sh <tag> -- the round-4 evidence set from ONE box (GPU box only; outputs under gpurun_out/<tag>/):
#   bench.json            python3 bench.py (the driver's command line)
#   kernel_stats*.csv     rocprofv3 --kernel-trace --stats of bench.py --serial (one un-overlapped trace launch per step: the
#                         duration the roofline is priced with) and of the default, pipelined command
#   pmc/                  counter passes of bench.py --serial (tools/pmc.sh) -> pmc.json (profiles/pmc_latest.json)
#   build_stats.csv       kernels of the device scene build (tools/scene_build_loop.py)
set -u
TAG=${1:-r4final}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o run -- python3 $R/bench.py --serial --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline --no-caller-path > $OUT/stats_serial.log 2>&1 || echo "serial stats run failed"
find $OUT/stats_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_serial.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipe -o run -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline --no-caller-path > $OUT/stats_pipe.log 2>&1 || echo "pipelined stats run failed"
find $OUT/stats_pipe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_build -o run -- python3 $R/tools/scene_build_loop.py > $OUT/stats_build.log 2>&1 || echo "build stats run failed"
find $OUT/stats_build -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/build_stats.csv
echo "stats done"
cd $R
tools/pmc.sh $TAG/pmc > /dev/null 2>&1
cp gpurun_out/$TAG/pmc/summary.txt $OUT/pmc_summary.txt 2>/dev/null
cp gpurun_out/$TAG/pmc/pmc.json $OUT/pmc.json 2>/dev/null
echo "pmc done"
head -3 $OUT/kernel_stats_serial.csv | cut -c1-160
python3 -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['serial_ms_per_step'], d['roofline']['kernel_ms'], d['cpu_baseline']['value'], d['config']['scene_create_ms'], d['config'].get('caller_path_ms'), d['config'].get('run_simulation_ms'))"

#!/bin/bash
# tools/round2_profile.sh <tag> -- the evidence set of round 2 from ONE box (GPU box only; outputs under gpurun_out/<tag>/):
# bench line, rocprofv3 kernel statistics of the same command, PMC passes of the trace kernel on the flat and on the
# rough scene and of the private-refill variant, the C4 timings, traversal statistics and trace times of both scenes.
set -u
TAG=${1:-r2prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caller-path > $OUT/stats.log 2>&1 || echo "stats run failed"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo "stats done"
cd $R
tools/pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; echo "pmc flat exit $?"
PMC_SCENE=synth_rough_A6 PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-caller-path --scene synth_rough_A6" tools/pmc.sh $TAG/pmc_rough > $OUT/pmc_rough.log 2>&1; echo "pmc rough exit $?"
LRC_REFILL=2 LRC_REFILL_W=7 tools/pmc.sh $TAG/pmc_refill trace_refill > $OUT/pmc_refill.log 2>&1; echo "pmc refill exit $?"
timeout -k 10 300 python3 tools/c4_time.py > $OUT/c4.json 2> $OUT/c4.err; echo "c4 exit $?"
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office synth_rough_A1; do
  timeout -k 10 120 python3 tools/trace_time.py $sc 2>/dev/null | tail -1 >> $OUT/trace_times.txt
  timeout -k 10 120 python3 tools/trav_stats.py $sc > $OUT/trav_$sc.txt 2>&1
done
timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-caller-path --scene synth_rough_A6 > $OUT/bench_rough.json 2>> $OUT/bench.err
cat $OUT/trace_times.txt
python3 -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('caller_path_rays_per_s'), d['config'].get('run_simulation_rays_per_s'))"

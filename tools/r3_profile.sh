#!/bin/bash
# round-3 evidence batch (GPU box): GPU suite, bench, rocprofv3 kernel statistics (bench + a scene build), PMC passes of
# the three scenes, virtual-world sweep, run_simulation / per-waypoint timings
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r3p
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q -s > $O/gputests.log 2>&1; echo "gpu tests exit $?" > $O/log.txt
tail -3 $O/gputests.log >> $O/log.txt
# counters first: bench.py derives roofline.frac from the counter file of THIS binary (source fingerprint)
tools/pmc.sh r3p/pmc > /dev/null 2>&1; cp gpurun_out/r3p/pmc/summary.txt $O/pmc_summary.txt; cp gpurun_out/r3p/pmc/pmc.json $O/pmc_synth_A6_office2.json
cp $O/pmc_synth_A6_office2.json profiles/pmc_latest.json; cp $O/pmc_synth_A6_office2.json profiles/pmc_synth_A6_office2.json
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err || { echo bench failed >> $O/log.txt; tail -5 $O/bench.err >> $O/log.txt; }
echo "bench done" >> $O/log.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1 || echo "stats run failed" >> $O/log.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/build_stats -o run -- python3 $R/tools/scene_build_loop.py > $O/build_stats.log 2>&1 || echo "build stats run failed" >> $O/log.txt
find $O/build_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/build_kernel_stats.csv
echo "stats done" >> $O/log.txt
cd $R
for sc in synth_rough_A6 synth_hall; do
  PMC_SCENE=$sc tools/pmc.sh r3p/pmc_$sc > /dev/null 2>&1; cp gpurun_out/r3p/pmc_$sc/summary.txt $O/pmc_${sc}_summary.txt; cp gpurun_out/r3p/pmc_$sc/pmc.json $O/pmc_$sc.json
  cp $O/pmc_$sc.json profiles/pmc_$sc.json
  timeout -k 10 200 python3 bench.py --scene $sc --no-cpu-baseline --no-caller-path > $O/bench_$sc.json 2>> $O/bench.err
done
echo "pmc done" >> $O/log.txt
VW_LIST="1 2 4 8" tools/vw_sweep.sh > $O/vw_sweep.txt 2>&1
cat $O/vw_sweep.txt >> $O/log.txt
timeout -k 10 200 python3 tools/run_sim_profile.py > $O/run_sim_profile.txt 2>&1
grep -E "run_simulation ms|per-waypoint" $O/run_sim_profile.txt >> $O/log.txt
timeout -k 10 200 python3 tools/per_waypoint_time.py >> $O/log.txt 2>&1
timeout -k 10 300 python3 tools/c4_time.py 256 > $O/c4.json 2>> $O/log.txt
head -8 $O/kernel_stats.csv | cut -c1-200 >> $O/log.txt
head -30 $O/build_kernel_stats.csv | cut -c1-200 >> $O/log.txt
python3 -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('scene_create_ms'), d['config'].get('run_simulation_ms'), d['config'].get('caller_path_ms'))" >> $O/log.txt

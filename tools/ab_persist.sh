#!/bin/bash
# tools/ab_persist.sh -- same-box A/B of the persistent trace launch against one workgroup per tile (laboratory build,
# LRC_PERSIST=0/1), HIP-event medians of the trace kernel alone, four scenes, three alternating repeats; then pose counts.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export LRC_LIB=$(ls $R/indoor*/liblidarcast_lab.so)
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall; do
  for rep in 1 2 3; do
    echo -n "one-shot   "; LRC_PERSIST=0 timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    echo -n "persistent "; LRC_PERSIST=1 timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
  done
done
for P in 16 32 128 256; do
  echo -n "one-shot   "; LRC_PERSIST=0 timeout -k 10 120 python3 tools/trace_time.py synth_A6_office2 32 2048 $P 2>&1 | tail -1
  echo -n "persistent "; LRC_PERSIST=1 timeout -k 10 120 python3 tools/trace_time.py synth_A6_office2 32 2048 $P 2>&1 | tail -1
done

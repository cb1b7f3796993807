#!/bin/bash
# round-3 measurement batch (GPU box): digests of the A/B builds, trace-kernel A/B, seeded-stream and C4 timings
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3
mkdir -p $O
echo "== digests" > $O/measure.log
for l in "" build_variants/edge.so; do
  echo -n "${l:-in-tree}: " >> $O/measure.log
  LRC_LIB=${l:+$R/$l} timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure.log
done
echo -n "edge_lab leaf pairs: " >> $O/measure.log
LRC_LIB=$R/build_variants/edge_lab.so LRC_LEAFW=2 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure.log
echo -n "edge_lab every 3rd redone: " >> $O/measure.log
LRC_LIB=$R/build_variants/edge_lab.so LRC_DEBUG_FORCE_REDO=3 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure.log
echo "== trace A/B (t,prim,point3,sem,ins,tile_count)" >> $O/measure.log
export LRC_TT_WANT=t,prim,point3,sem,ins,tile_count
P=$R/indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office; do
  for rep in 1 2 3; do
    echo -n "in-tree " >> $O/measure.log; timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure.log
    echo -n "edge " >> $O/measure.log; LRC_LIB=$R/build_variants/edge.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure.log
    echo -n "lab leafw2 " >> $O/measure.log; LRC_LIB=$P/liblidarcast_lab.so LRC_LEAFW=2 timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure.log
    echo -n "edge leafw2 " >> $O/measure.log; LRC_LIB=$R/build_variants/edge_lab.so LRC_LEAFW=2 timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure.log
  done
done
echo "== seeded stream" >> $O/measure.log
timeout -k 10 300 python3 tools/rng_time.py 256 >> $O/measure.log 2>&1
echo "== C4" >> $O/measure.log
timeout -k 10 400 python3 tools/c4_time.py 256 >> $O/measure.log 2>&1
echo "== per waypoint" >> $O/measure.log
timeout -k 10 200 python3 tools/per_waypoint_time.py >> $O/measure.log 2>&1
echo "== run_simulation profile" >> $O/measure.log
timeout -k 10 200 python3 tools/run_sim_profile.py >> $O/measure.log 2>&1
echo "== bench" >> $O/measure.log
timeout -k 10 300 python3 bench.py > $O/bench1.json 2>> $O/measure.log
tail -c 600 $O/bench1.json >> $O/measure.log

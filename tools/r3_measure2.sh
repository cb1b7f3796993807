#!/bin/bash
# round-3 measurement batch 2 (GPU box): GPU suite, digests and trace A/B of the edge-record builds, per-waypoint loop
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r3
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -s > $O/gputests2.log 2>&1; echo "gpu tests exit $?" > $O/measure2.log
tail -4 $O/gputests2.log >> $O/measure2.log
echo "== digests" >> $O/measure2.log
for l in "" build_variants/edge.so; do
  echo -n "${l:-in-tree}: " >> $O/measure2.log
  LRC_LIB=${l:+$R/$l} timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
done
echo -n "edge host builder: " >> $O/measure2.log
LRC_LIB=$R/build_variants/edge.so LRC_DEVICE_BUILD=0 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
echo -n "edge float32 nodes: " >> $O/measure2.log
LRC_LIB=$R/build_variants/edge.so LRC_QNODES=0 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
echo -n "edge_lab leaf pairs: " >> $O/measure2.log
LRC_LIB=$R/build_variants/edge_lab.so LRC_LEAFW=2 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
echo -n "edge_lab every 3rd redone: " >> $O/measure2.log
LRC_LIB=$R/build_variants/edge_lab.so LRC_DEBUG_FORCE_REDO=3 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
echo -n "edge_lab every ray redone: " >> $O/measure2.log
LRC_LIB=$R/build_variants/edge_lab.so LRC_DEBUG_FORCE_REDO=1 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1 >> $O/measure2.log
echo "== trace A/B (t,prim,point3,sem,ins,tile_count)" >> $O/measure2.log
export LRC_TT_WANT=t,prim,point3,sem,ins,tile_count
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall; do
  for rep in 1 2 3; do
    echo -n "in-tree " >> $O/measure2.log; timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure2.log
    echo -n "edge " >> $O/measure2.log; LRC_LIB=$R/build_variants/edge.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure2.log
    echo -n "edge leafw2 " >> $O/measure2.log; LRC_LIB=$R/build_variants/edge_lab.so LRC_LEAFW=2 timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1 >> $O/measure2.log
  done
done
echo "== full records (t,prim,normal3,point3,sem,ins,tile_count)" >> $O/measure2.log
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for rep in 1 2; do
  echo -n "in-tree " >> $O/measure2.log; timeout -k 10 120 python3 tools/trace_time.py synth_A6_office2 2>&1 | tail -1 >> $O/measure2.log
  echo -n "edge " >> $O/measure2.log; LRC_LIB=$R/build_variants/edge.so timeout -k 10 120 python3 tools/trace_time.py synth_A6_office2 2>&1 | tail -1 >> $O/measure2.log
done
echo "== per waypoint" >> $O/measure2.log
timeout -k 10 200 python3 tools/per_waypoint_time.py >> $O/measure2.log 2>&1
timeout -k 10 200 python3 tools/run_sim_profile.py 2>&1 | grep -E "run_simulation ms|per-waypoint" >> $O/measure2.log

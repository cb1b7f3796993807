#!/bin/bash
# A/B of the dead-pop skip (build_variants/deadpop_lab.so) against the in-tree library: digests, then trace times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
P=$R/indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd
echo -n "in-tree digest: "; timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
echo -n "deadpop digest: "; LRC_LIB=$R/build_variants/deadpop_lab.so LRC_SECTOR=0 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
echo -n "deadpop, float32 nodes digest: "; LRC_LIB=$R/build_variants/deadpop_lab.so LRC_SECTOR=0 LRC_QNODES=0 timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall; do
  for rep in 1 2 3; do
    echo -n "in-tree "; timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    echo -n "deadpop "; LRC_LIB=$R/build_variants/deadpop_lab.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    echo -n "deadpop-off(same binary) "; LRC_DEADPOP_OFF=1 LRC_LIB=$R/build_variants/deadpop_lab.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
  done
done

#!/bin/bash
R=${GRAFT_REPO_ROOT}
cd $R
echo -n "in-tree digest: "; timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
echo -n "variant digest: "; LRC_LIB=$R/build_variants/$1.so timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall; do
  for rep in 1 2 3; do
    echo -n "in-tree "; timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    echo -n "$1 "; LRC_LIB=$R/build_variants/$1.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
  done
done

#!/bin/bash
# tools/ab_time.sh [scene ...] -- same-box A/B of the trace kernel: the in-tree library against every library in
# build_variants/ (LRC_LIB), alternating, trace_time.py medians.  GPU box only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export LRC_TT_WANT=${LRC_TT_WANT:-t,prim,point3,sem,ins,tile_count}
SCENES=${@:-synth_A6_office2}
for sc in $SCENES; do
  for rep in 1 2; do
    for l in "" $(ls build_variants/*.so 2>/dev/null); do
      echo -n "${l:-in-tree} "
      LRC_LIB=${l:+$R/$l} timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    done
  done
done

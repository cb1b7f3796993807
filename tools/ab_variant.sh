#!/bin/bash
# tools/ab_variant.sh <name> [<name> ...] -- same-box A/B of build_variants/<name>.so against the in-tree library: the
# digest of tools/variant_digest.py must be equal, then the trace kernel's HIP-event medians on four scenes, alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
echo -n "in-tree digest: "; timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
for v in "$@"; do
  echo -n "$v digest: "; LRC_LIB=$R/build_variants/$v.so timeout -k 10 300 python3 tools/variant_digest.py 2>&1 | tail -1
done
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in ${AB_SCENES:-synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall}; do
  for rep in 1 2 3; do
    echo -n "in-tree "; timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    for v in "$@"; do
      echo -n "$v "; LRC_LIB=$R/build_variants/$v.so timeout -k 10 120 python3 tools/trace_time.py $sc 2>&1 | tail -1
    done
  done
done

#!/bin/bash
# tools/ab_env.sh "<label>:<ENV=V ENV2=V ...>" ... -- same-box A/B of environment knobs of the laboratory build
# (liblidarcast_lab.so through LRC_LIB): HIP-event medians of the trace kernel alone, four scenes, three alternating repeats.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export LRC_LIB=$(ls $R/indoor*/liblidarcast_lab.so)
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in ${AB_SCENES:-synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall}; do
  for rep in 1 2 3; do
    for v in "$@"; do
      label=${v%%:*}; envs=${v#*:}
      echo -n "$label "; env $envs timeout -k 10 120 python3 tools/trace_time.py $sc ${AB_ARGS:-} 2>&1 | tail -1
    done
  done
done

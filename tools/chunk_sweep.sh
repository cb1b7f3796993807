#!/bin/bash
# tools/chunk_sweep.sh -- trace kernel with the tiles dealt to the XCDs in chunks (LRC_TILE_CHUNK tiles, round robin)
# against the default (eight contiguous ranges), over sensor shapes and scenes, alternating.  GPU box only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export LRC_TT_WANT=t,prim,normal3,point3,sem,ins,tile_count
for sc in ${SCENES:-synth_A6_office2 synth_rough_A6 synth_A1_office synth_hall}; do
  for shape in "32 2048 64" "8 512 256" "16 1024 128" "64 4096 16" "32 2048 8" "32 4000 32" "8 512 16"; do
    set -- $shape
    for rep in 1 2 3; do
      for c in ${CHUNKS:-0 64 128}; do
        echo -n "chunk=$c " ; LRC_LIB=$R/indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd/liblidarcast_lab.so LRC_TILE_CHUNK=$c timeout -k 10 120 python3 tools/trace_time.py $sc $1 $2 $3 2>&1 | tail -1
      done
    done
  done
done

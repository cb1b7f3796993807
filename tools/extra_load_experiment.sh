#!/bin/bash
# tools/extra_load_experiment.sh build | run -- what prices a divergent node step of the float32-node trace kernel.
# build (any box with hipcc): variant libraries build_variants/libx{1,2,3}.so with N extra 16-byte loads of the node's own
# cache line per divergent node step (-DLRC_EXP_EXTRA_NODE_LOADS=N).  run (GPU box): trace time of the default and the
# variants with the float32 nodes (LRC_QNODES=0).  Result of round 2: profiles/r02_extra_load_experiment.txt.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PKG=$(ls -d $R/indoor-point-cloud-*_amd)
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -Xarch_device -fno-honor-nans -Xarch_device -fno-slp-vectorize -fPIC -shared -std=c++17 -pthread"
if [ "${1:-run}" = build ]; then
  mkdir -p $R/build_variants
  cd $PKG/csrc
  for x in 1 2 3; do
    hipcc $FLAGS -DLRC_EXP_EXTRA_NODE_LOADS=$x lidarcast.hip lrc_nn.hip lrc_metrics.hip lrc_occupancy.hip bvh_build.cpp lrc_qnodes.cpp -o $R/build_variants/libx$x.so || exit 1
  done
  exit 0
fi
cd $R
export LRC_QNODES=0 LRC_TT_WANT=t,prim,point3,sem,ins,tile_count
for l in "" build_variants/libx1.so build_variants/libx2.so build_variants/libx3.so ""; do
  echo -n "N=${l:+${l//[^0-9]/}} "
  LRC_LIB=${l:+$R/$l} timeout -k 10 120 python3 tools/trace_time.py 2>&1 | tail -1
done

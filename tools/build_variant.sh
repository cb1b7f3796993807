#!/bin/bash
# tools/build_variant.sh <name> [extra compiler flags ...] -- an A/B build of the whole library into build_variants/<name>.so
# (loaded through LRC_LIB by tools/ab_time.sh, tools/variant_digest.py).  Cross-compiles without a GPU.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd
NAME=$1; shift
mkdir -p $R/build_variants/obj_$NAME
cd $P/csrc
HIPF="--offload-arch=gfx950 -O3 -ffp-contract=off -Xarch_device -fno-honor-nans -Xarch_device -fno-slp-vectorize -fPIC -std=c++17 -pthread"
pids=()
for f in lidarcast.hip lrc_bvh_device.hip lrc_nn.hip lrc_metrics.hip lrc_occupancy.hip; do
  /opt/rocm/bin/hipcc $HIPF "$@" -c $f -o $R/build_variants/obj_$NAME/${f%.*}.o & pids+=($!)
done
for f in bvh_build.cpp lrc_qnodes.cpp lrc_nprandom.cpp; do
  g++ -O3 -ffp-contract=off -fPIC -std=c++17 -pthread "$@" -c $f -o $R/build_variants/obj_$NAME/${f%.*}.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $R/build_variants/obj_$NAME/*.o -o $R/build_variants/$NAME.so
rm -rf $R/build_variants/obj_$NAME
echo built $R/build_variants/$NAME.so

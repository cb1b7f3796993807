#!/bin/bash
# tools/chunk_scheme_sweep.sh -- the pose chunks of the frame pipeline (trace + compaction of chunk c+1 beside the transfer of
# chunk c): chunk ends in 32nds of the trajectory, laboratory build (LRC_CHUNK_SCHEME), C3 through lrc_scan_poses_compact.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export LRC_LIB=$(ls $R/indoor*/liblidarcast_lab.so)
for rep in 1 2; do
  for sch in 0 "2,8,16,32" "1,4,16,32" "2,6,16,32" "1,3,8,16,32" "2,8,32" "4,16,32" "1,4,12,32" "2,6,14,22,32" "1,2,4,8,16,32"; do
    echo -n "scheme $sch: "; LRC_CHUNK_SCHEME=$sch timeout -k 10 120 python3 tools/compact_time.py 2>&1 | grep "('point3', 'sem', 'ins') "
  done
done

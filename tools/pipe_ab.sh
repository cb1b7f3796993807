#!/bin/bash
# tools/pipe_ab.sh <name> ... -- same-box A/B of build_variants/<name>.so against the in-tree library on tools/pipe_time.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  echo "in-tree: $(PIPE_STEPS=200 timeout -k 10 120 python3 tools/pipe_time.py $PIPE_SCENE 2>&1 | grep "ms/step" | tail -1)"
  for v in "$@"; do
    echo "$v: $(LRC_LIB=$R/build_variants/$v.so PIPE_STEPS=200 timeout -k 10 120 python3 tools/pipe_time.py $PIPE_SCENE 2>&1 | grep "ms/step" | tail -1)"
  done
done

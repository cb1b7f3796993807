#!/bin/bash
# tools/bench_ab.sh <name> ... -- same-box A/B of build_variants/<name>.so against the in-tree library on the bench's own timed
# blocks (python3 bench.py --steps 20 --warmup 5, host legs off): ms_per_step of the 20-step blocks, alternating, three rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
one() {
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --min-seconds ${AB_SECONDS:-2} --no-cpu-baseline --no-caller-path 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  serial %.4f  value %.3f G' % (d['ms_per_step'], d['config']['serial_ms_per_step'], d['value'] / 1e9))"
}
for rep in 1 2 3; do
  echo "in-tree: $(one)"
  for v in "$@"; do echo "$v: $(LRC_LIB=$R/build_variants/$v.so one)"; done
done

#!/bin/bash
# tools/vw_sweep.sh -- single-GPU diagnostics of the N-GPU step: bench.py --dist-selftest --virtual-world W
# for the default library and the variant libraries in build_variants/ (LRC_LIB).  GPU box only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/vw
cd $R
run() {  # tag, W, extra env
  local tag=$1 W=$2
  timeout -k 10 200 python3 bench.py --dist-selftest --virtual-world $W --steps 50 --warmup 5 --no-cpu-baseline \
      > gpurun_out/vw/$tag.json 2> gpurun_out/vw/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/vw/$tag.err; return 1; }
  python3 - <<PY
import json
d = json.loads(open('gpurun_out/vw/$tag.json').read().strip().splitlines()[-1])
c = d['config']
print('$tag', 'W=$W', 'ms_per_step=%.4f' % d['ms_per_step'], 'trace_ms=%.4f' % d['roofline']['kernel_ms'],
      'payload', c.get('gather_payload'), 'calibration', {k: round(v, 4) for k, v in (c.get('gather_payload_calibration_ms_per_step') or {}).items()})
PY
}
for W in ${VW_LIST:-8}; do
  run default_W$W $W || exit 1
  for lib in build_variants/liblidarcast_*.so; do
    [ -f "$lib" ] || continue
    tag=$(basename $lib .so | sed s/liblidarcast_//)
    LRC_LIB=$R/$lib run ${tag}_W$W $W || exit 1
  done
done

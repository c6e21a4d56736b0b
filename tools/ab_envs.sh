#!/bin/bash
# where does the split-bf16 path start to win?  per build flag set, the default bench at several batch sizes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for flags in "$@"; do
  MEL_HIPCC_FLAGS="$flags" python -m melissa_amd.build --force > gpurun_out/ab_envs_build.log 2>&1 || { echo "build failed: $flags"; continue; }
  for envs in ${ENVS:-128 256 384 512 768}; do
    MEL_HIPCC_FLAGS="$flags" python bench.py --envs $envs ${AB_ARGS} --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$flags envs=$envs', round(d['value']/1e6, 3), 'M/s', round(d['ms_per_step'], 4), 'ms')
"
  done
done
MEL_HIPCC_FLAGS="" python -m melissa_amd.build --force > /dev/null 2>&1

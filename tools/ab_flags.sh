#!/bin/bash
# A/B of build flags on the GPU box: for every flag set rebuild the library and run the bench (sustained leg only).
#   bash tools/ab_flags.sh "<flags A>" "<flags B>" ...      ("" = the default build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/ab
i=0
for flags in "$@"; do
  i=$((i+1))
  MEL_HIPCC_FLAGS="$flags" python -m melissa_amd.build --force > gpurun_out/ab/build_$i.log 2>&1 || { echo "build failed: $flags"; tail -5 gpurun_out/ab/build_$i.log; continue; }
  MEL_HIPCC_FLAGS="$flags" python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-extra-legs --no-profile ${AB_ARGS} > gpurun_out/ab/line_$i.json 2> gpurun_out/ab/err_$i.log
  MEL_HIPCC_FLAGS="$flags" python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/ab/line_$i.json") if l.startswith("{")][0])
    print("[$i] flags=%r value=%.3f M/s ms_per_step=%.4f errors=%s" % ("$flags", d["value"] / 1e6, d["ms_per_step"], d["config"]["env_error_flags"]))
except Exception as e:
    print("[$i] flags=%r FAILED %r" % ("$flags", e))
PY
done
MEL_HIPCC_FLAGS="" python -m melissa_amd.build --force > /dev/null 2>&1

#!/bin/bash
# A/B of build flags on the GPU box with the stage timer: bash tools/ab_stage.sh "<flags A>" "<flags B>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/ab
i=0
for flags in "$@"; do
  i=$((i+1))
  MEL_HIPCC_FLAGS="$flags" python -m melissa_amd.build --force > gpurun_out/ab/sbuild_$i.log 2>&1 || { echo "build failed: $flags"; tail -5 gpurun_out/ab/sbuild_$i.log; continue; }
  MEL_HIPCC_FLAGS="$flags" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs ${AB_ARGS} > gpurun_out/ab/sline_$i.json 2> gpurun_out/ab/serr_$i.log
  MEL_HIPCC_FLAGS="$flags" python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/ab/sline_$i.json") if l.startswith("{")][0])
    print("[$i] flags=%r value=%.3f M/s ms=%.4f frac=%.3f errors=%s" % ("$flags", d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d["config"]["env_error_flags"]))
    print("     ", d["stage_us"])
except Exception as e:
    print("[$i] flags=%r FAILED %r" % ("$flags", e))
PY
done
MEL_HIPCC_FLAGS="" python -m melissa_amd.build --force > /dev/null 2>&1

#!/bin/bash
# A/B of build flags on the GPU box over tools/gemm_bench.py: AB_ARGS="--split --shapes big" bash tools/ab_gemm.sh "<flags A>" "<flags B>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/abg
i=0
for flags in "$@"; do
  i=$((i+1))
  MEL_HIPCC_FLAGS="$flags" python -m melissa_amd.build --force > gpurun_out/abg/build_$i.log 2>&1 || { echo "build failed: $flags"; tail -5 gpurun_out/abg/build_$i.log; continue; }
  echo "[$i] flags='$flags'"
  MEL_HIPCC_FLAGS="$flags" timeout -k 10 300 python tools/gemm_bench.py --rounds 8 ${AB_ARGS} 2>&1 | grep -v amdgpu.ids
done
MEL_HIPCC_FLAGS="" python -m melissa_amd.build --force > /dev/null 2>&1

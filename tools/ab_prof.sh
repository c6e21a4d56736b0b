#!/bin/bash
# build with the split-kernel cycle stamps, run tools/split_prof.py on the given shapes, rebuild the default library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
MEL_HIPCC_FLAGS="-DMEL_GEMM_PROF=99 -DMEL_SPLIT_PROF $EXTRA" python -m melissa_amd.build --force > gpurun_out/prof_build.log 2>&1 || { tail -5 gpurun_out/prof_build.log; exit 1; }
export MEL_HIPCC_FLAGS="-DMEL_GEMM_PROF=99 -DMEL_SPLIT_PROF $EXTRA"
for shape in "$@"; do timeout -k 10 120 python tools/split_prof.py $shape 2>&1 | grep -v amdgpu.ids; done
MEL_HIPCC_FLAGS="" python -m melissa_amd.build --force > /dev/null 2>&1

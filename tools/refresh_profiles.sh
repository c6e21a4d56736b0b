#!/bin/bash
# Profile refresh (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the bench command, the two PMC
# passes behind roofline.traffic (tied to the library's source hash), the bench lines, the soak-parity log.
# Usage: bash tools/refresh_profiles.sh r02b        (outputs land in gpurun_out/<tag>/ and are copied into profiles/)
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# same command as the default bench (HIP-graph replay, episode stream), minus the legs and the CPU baseline
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.log
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph > /dev/null 2> $OUT/pmc_fetch.log
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph > /dev/null 2> $OUT/pmc_write.log
echo "pmc write done"
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs   (MI355X, round 3, round-batched loop, fp32 results with the precision chosen per launch - MEL_PREC_F32_AUTO -, HIP-graph replay, device episode stream)";
  echo "# 942 launches per forward kernel = 512 untimed settling rounds + 30 warm-up + 200 timed (four rounds per graph replay) + 200 stage-timer steps (the stage timer's eager pass launches the same kernels); episode_* = the episode stream's refill on its side stream";
  cat $(find $OUT/stats -name '*kernel_stats.csv' | head -1); } > $OUT/${TAG}_round_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json       # bench.py reads roofline.traffic from profiles/ (hash-checked)
cp $OUT/${TAG}_round_kernel_stats.csv profiles/${TAG}_round_kernel_stats.csv   # ... and profiler_avg_launch_us from the summary beside it
python3 bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.log
echo "bench line done"
if [ -z "$NO_SOAK" ]; then       # NO_SOAK=1: the soak (15-20 min with the 100-node cases) runs in its own gpurun call
  python3 tools/soak_parity.py ldgn > $OUT/${TAG}_soak_parity.log 2>&1 || echo "SOAK FAILED"
  tail -2 $OUT/${TAG}_soak_parity.log
fi
tail -c 400 $OUT/${TAG}_bench_line.json

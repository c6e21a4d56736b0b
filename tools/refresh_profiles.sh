#!/bin/bash
# Round-end profile refresh (run on the GPU box through gpurun): kernel-trace stats of the bench command, the two
# PMC passes behind roofline.traffic, and the bench line itself.  Usage: bash tools/refresh_profiles.sh r01k
set -e
TAG=${1:-r01k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs --no-graph > $OUT/bench_line_nograph.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-profile --no-graph > /dev/null 2> $OUT/pmc_write.log
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_round_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json       # bench.py reads roofline.traffic from profiles/
python3 bench.py > $OUT/${TAG}_bench_line_round.json 2> $OUT/bench.log
python3 bench.py --dtype bf16 --no-cpu-baseline > $OUT/${TAG}_bench_line_round_bf16.json 2>> $OUT/bench.log
python3 bench.py --dtype f32s --no-cpu-baseline > $OUT/${TAG}_bench_line_round_f32s.json 2>> $OUT/bench.log
tail -c 600 $OUT/${TAG}_bench_line_round.json

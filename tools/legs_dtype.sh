# f32 vs f32s vs f32a (per-launch choice) over the bench's configurations: bash tools/legs_dtype.sh
for args in "" "--model hl_dgn --envs 512" "--nodes 20 --envs 256" "--nodes 100" "--model dgn_r" "--mode aec"; do
  for dt in f32 f32s f32a; do
    python bench.py --dtype $dt $args --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs --no-profile 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$dt $args', round(d['value']/1e6, 3), 'M/s', round(d['ms_per_step'], 4), 'ms', d['config']['env_error_flags'])
"
  done
done

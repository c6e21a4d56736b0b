#!/bin/bash
for g in "" "--no-graph"; do
for s in 1 2; do
  timeout -k 10 200 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-profile --streams $s $g 2>/dev/null | tail -1 > /tmp/line.json
  python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('streams', $s, '$g', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms/step')"
done
done

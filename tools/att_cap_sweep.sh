#!/bin/bash
# tuning aid: sweep the attention kernels' LDS row cap (MEL_ATT_CAP) and print the stage timers
for c in 0 8 16 32; do
  MEL_ATT_CAP=$c timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python3 -c "
import json; d=json.load(open('/tmp/line.json'))
print('cap', $c, round(d['value']/1e6,2), 'M/s', {k:d['stage_us'][k] for k in ('conv1_att','conv2_att')})"
done

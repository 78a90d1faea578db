#!/bin/bash
# The three mesh configurations at 1080p (c3, c5: 64 spp; c4: 16 spp) through bench.py: device ms per frame and the quantum preset.
for c in c3 c5 c4; do spp=64; [ $c = c4 ] && spp=16
  timeout -k 10 300 python bench.py --config $c --size 1920x1080 --spp $spp --steps ${1:-3} --warmup 1 --no-cpu-baseline \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step_device_only'], 'ms', d['value'], 'Mrays/s', 'walk_tune', d['config'].get('walk_tune'))"
done

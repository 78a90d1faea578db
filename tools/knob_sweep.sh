#!/bin/bash
# Traversal quantum sweep of the mesh kernel on one configuration at 1080p: tools/knob_sweep.sh c4 16 "8 48 2" "4 48 2" ...
cfg=$1; spp=$2; shift 2
for k in "$@"; do set -- $k
  SRT_WALK_ROUNDS=$1 SRT_WALK_VOTE=$2 SRT_WALK_STEPS=$3 timeout -k 10 200 python bench.py --config $cfg --size 1920x1080 --spp $spp --steps 2 --warmup 1 --no-cpu-baseline \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg rounds=$1 vote=$2 steps=$3 ms', d['ms_per_step_device_only'])"
done

# Mesh-kernel knobs on the 983,040-triangle scene (64 spp): ms per frame for each setting
run() { echo "$* $(env $* timeout -k 10 200 python tools/big_scenes.py c5 --spp 64 --once 2>/dev/null | grep config)"; }
run SRT_WALK_STEPS=1
run SRT_WALK_STEPS=2
run SRT_WALK_STEPS=3
run SRT_WALK_STEPS=6
run SRT_WALK_STEPS=2 SRT_WALK_VOTE=8
run SRT_WALK_STEPS=2 SRT_WALK_VOTE=32

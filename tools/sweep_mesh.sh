# Mesh-kernel knobs on the 983,040-triangle scene (64 spp): ms per frame for each setting.
#   bash tools/sweep_mesh.sh        (on a GPU box, from the repository root)
run() { echo "$* $(env $* timeout -k 10 200 python tools/big_scenes.py c5 --spp 64 --once 2>/dev/null | grep config)"; }
run SRT_WALK_STEPS=2
for v in 1 3 4 6; do run SRT_WALK_STEPS=$v; done
for v in 8 32 48; do run SRT_WALK_VOTE=$v; done
for v in 2 8 32; do run SRT_WALK_ROUNDS=$v; done

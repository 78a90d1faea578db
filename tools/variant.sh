#!/bin/bash
# usage: bash tools/variant.sh <name> [extra compiler flags...]
# Builds the library with extra flags (e.g. -DSRT_SPEC_LEAF=0) into tools/_variants/<name>/libsrt.so; experiments load
# it with SRT_LIBRARY=tools/_variants/<name>/libsrt.so.  Built on the CPU box; the .so travels to the GPU box.
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/simple_raytracer_amd/csrc
mkdir -p $R/tools/_variants/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-parameter -Wno-unused-value "$@" -shared -o $R/tools/_variants/$N/libsrt.so \
  $S/srt_kernels.hip $S/srt_wavefront.hip $S/srt_post.hip -x hip $S/srt_api.cpp $S/srt_scene.cpp -ldl
echo "built tools/_variants/$N/libsrt.so"

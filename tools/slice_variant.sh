#!/bin/bash
# A variant of the library whose bit-sliced kernel is compiled with extra -D flags (A/B runs on ONE GPU box: boxes differ by 1-2 %):
#   tools/slice_variant.sh <tag> [-DPA_SLICE_...]   -> astar-pairwise-aligner_amd/libastarpa_c_hip_<tag>.so   (PA_LIB_PATH selects it)
# The other units come from the product build's objects (build the product first).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
C=$R/astar-pairwise-aligner_amd/csrc; B=$R/astar-pairwise-aligner_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include "$@" -c $C/slice_unit.hip -o $B/slice_unit_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/astar-pairwise-aligner_amd/libastarpa_c_hip_$TAG.so $B/slice_unit_$TAG.o $(ls $B/*.o | grep -v "slice_unit\|engine_hip_timers")
ls -la $R/astar-pairwise-aligner_amd/libastarpa_c_hip_$TAG.so

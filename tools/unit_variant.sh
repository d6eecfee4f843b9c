#!/bin/bash
# A variant of the library with ONE translation unit compiled with extra -D flags (A/B runs on one GPU box):
#   tools/unit_variant.sh <unit without .hip> <tag> [-D...]   -> astar-pairwise-aligner_amd/libastarpa_c_hip_<tag>.so   (PA_LIB_PATH selects it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
UNIT=$1; TAG=$2; shift; shift
C=$R/astar-pairwise-aligner_amd/csrc; B=$R/astar-pairwise-aligner_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include "$@" -c $C/$UNIT.hip -o $B/${UNIT}_var_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/astar-pairwise-aligner_amd/libastarpa_c_hip_$TAG.so $B/${UNIT}_var_$TAG.o $(ls $B/*.o | grep -v "/${UNIT}\.o\|_var_\|slice_unit_\|engine_hip_timers")
ls -la $R/astar-pairwise-aligner_amd/libastarpa_c_hip_$TAG.so

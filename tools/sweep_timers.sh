#!/bin/bash
# Where does a strip of the device-side sweep spend its time?  Builds a SECOND copy of the library with the phase clocks compiled in
# (-DPA_SWEEP_PHASE_TIMERS; the product build has none: they cost 16 scalar registers) and, on a GPU box, runs C3 through it one pass
# after the other, printing per pass: ms after its launch and the strip-microseconds per phase (begin / slow / cross / probes / end /
# bottom / plain / granule waits).
#   here (no GPU):   tools/sweep_timers.sh build        -> astar-pairwise-aligner_amd/libastarpa_c_hip_timers.so (git-ignored, travels with gpurun)
#   on the GPU box:  tools/sweep_timers.sh run [depth]  (default depth 1; 5 = pipelined as in the product)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
LIB=$R/astar-pairwise-aligner_amd/libastarpa_c_hip_timers.so
case "${1:-build}" in
  build)
    C=$R/astar-pairwise-aligner_amd/csrc
    B=$R/astar-pairwise-aligner_amd/build   # the product build's objects (python -c "import __graft_entry__ as g; g.build()" first)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPA_SWEEP_PHASE_TIMERS -I $R/include -c $C/engine_hip.hip -o $B/engine_hip_timers.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB $B/engine_hip_timers.o $(ls $B/*.o | grep -v "engine_hip")
    ls -la $LIB ;;
  run)
    [ -f $LIB ] || { echo "build it first (tools/sweep_timers.sh build)"; exit 2; }
    PA_LIB_PATH=$LIB PA_SWEEP_TIMING=1 PA_SWEEP_DEPTH=${2:-1} python $R/tools/c3_quick.py ;;
  *) echo "usage: $0 build | run [depth]"; exit 2 ;;
esac

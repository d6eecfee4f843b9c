"""astar-pairwise-aligner_amd -- MI355X-native drop-in for the bit-parallel block-DP hot path of
RagnarGrootKoerkamp/astar-pairwise-aligner (pa-bitpacking + the astarpa2 block engine).

The compute path is hand-written HIP for gfx950 in libastarpa_c_hip.so (built in-tree from csrc/);
this package is the thin host-side mirror of the reference's interfaces.  No CPU fallback exists.
"""
from . import _build, aligner, capi, generate  # noqa: F401
from .aligner import (AstarPa2, AstarPa2Params, BlockParams, astarpa2_full, astarpa2_nw,  # noqa: F401
                      astarpa2_simple, c_abi_align)
from .capi import (Batch, OperatorContext, PaError, align_file, align_multi, compute, fill, profile_build, read_pairs, require_gpu, search,  # noqa: F401
                   search_trace)


def align_batch(pairs):
    """[(cost, CIGAR)] of many independent pairs, forward pass and traceback both on the GPU (pa_batch_align).
    Same alignments as `AstarPa2Params.nw()` with sparse blocks (`pa_params_batch_align`)."""
    from .sharding import default_align

    return default_align(list(pairs))

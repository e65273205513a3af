"""crass_amd — MI355X-native (gfx950) engine for crass's DR search + read recruitment path.

The product is ``libcrass_hip.so`` (hand-written HIP kernels + C-ABI host engine, see
``include/crass_hip.h``); this package is the thin Python host layer used by tests, bench.py
and the multi-GPU driver.  There is no CPU fallback: importing works everywhere, but every
search entry point needs the compiled library and a GPU.
"""
from ._abi import LIB_PATH, SYMBOLS, load  # noqa: F401
from .engine import (ConsensusResult, CrassError, FastxFile, FastxIndex, PackedReads, SearchEngine, SearchGroup, default_params,  # noqa: F401
                     consensus, dr_slots, merge_host, merge_rebuild, build_outputs, search_pipeline, search_pipeline_group, stream_fastx, synth_packed, synth_spec, unpack_ascii)

__all__ = ["ConsensusResult", "consensus", "CrassError", "FastxFile", "FastxIndex", "PackedReads", "SearchEngine", "SearchGroup", "default_params", "search_pipeline", "search_pipeline_group", "merge_host", "merge_rebuild", "build_outputs", "dr_slots",
           "stream_fastx", "synth_packed", "synth_spec", "unpack_ascii", "load", "LIB_PATH", "SYMBOLS"]

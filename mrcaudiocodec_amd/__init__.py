"""
mrcaudiocodec_amd -- MI355X (gfx950) implementation of the per-block encode hot path of
laser55/mrcAudioCodec behind the reference's own `codecThem` function signatures.

    import mrcaudiocodec_amd.codecThem as codec      # drop-in for `import codecThem as codec` (pacfileThem.py:108)

Layout: csrc/ (HIP kernels + C ABI, built into libmrc_hip.so), _lib.py (ctypes binding),
codecThem.py (the reference's per-block interface), batch.py (device-resident batch / stream API used by
bench.py and the multi-GPU sharding), synth.py (synthetic PCM of BASELINE.md's configs).
Importing this package needs the built shared library; running anything needs a gfx950 GPU.
"""
from ._lib import Handle, ChainSchedule, MrcError, PinnedArray, LIB_PATH  # noqa: F401

__all__ = ["Handle", "ChainSchedule", "MrcError", "PinnedArray", "LIB_PATH"]

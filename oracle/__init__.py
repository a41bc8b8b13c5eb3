"""
oracle/ -- CPU restatement (Python 3 + NumPy, float64) of the per-block ENCODE path of
laser55/mrcAudioCodec.  THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it, and only
as the checker / the timed CPU baseline.  The product path (mrcaudiocodec_amd/) never imports it
and has no CPU fallback.

What it restates (reference file:line, relative to the reference checkout):
    window.py:28-45,49-121      -> oracle.window    (HanningWindow, KBDWindow, TransitionWindow)
    mdct.py:13-96               -> oracle.mdct      (MDCTslow, MDCT forward)
    psychoac.py:8-219           -> oracle.psychoac  (SPL, Intensity, Thresh, Bark, Masker,
                                                     band tables, getMaskedThreshold, CalcSMRs)
    bitalloc.py:106-155         -> oracle.bitalloc  (BitAlloc)
    quantize.py:12-38,61-87,114-146,222-249,294-322 -> oracle.quantize
    ms_stereo.py:5-27,53-81     -> oracle.ms_stereo
    codecThem.py:136-354,359-574-> oracle.codec     (EncodeSingleChannel, JointEncodeChannels,
                                                     Encode, EncodeNoHuff, JointEncode, Huffman gain)
    pacfileThem.py:586-660,793-830 (block framing, band-table choice) -> oracle.framing

The reference is Python 2 / legacy NumPy and cannot execute in this image (no python2; its
mdct/psychoac/codecThem/pacfileThem are SyntaxErrors under Python 3).  Every place where
Python-2 semantics change a result (integer `/`, float used as a size, int() truncation) is
restated explicitly and marked `py2:` in the code.

PARITY PINNING STATUS
    pinned by golden vectors generated here by importing the reference's own py3-importable
    modules (tests/golden/make_golden.py):  quantize.* , bitalloc.BitAlloc, ms_stereo.*,
    window.HanningWindow.
    pinned by the reference's own stated relations: MDCT == MDCTslow (mdct.py:185-199), TDAC
    known-answer vector (mdct.py:131-182).
    PARITY UNPINNED (no reference test, module not importable): KBDWindow / TransitionWindow,
    getMaskedThreshold / CalcSMRs, band tables, the orchestration in codecThem.py, Huffman gain.
    For those the restatement below *is* the specification the HIP path is checked against.

Two flavours with identical results:
    oracle.<module>   "faithful": one block at a time, same redundancy as the reference (per-call
                      window construction, duplicate masked-threshold evaluation, Python peak
                      loop).  This is what bench.py times as the CPU baseline (kind="port").
    oracle.fast       batched / vectorised; same arithmetic, same summation orders; used by the
                      parity tests so 10^3..10^4 frames finish in seconds.
"""

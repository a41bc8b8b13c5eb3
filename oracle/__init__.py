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

The reference is Python 2 / legacy NumPy: there is no python2 in this image and, imported plainly under
Python 3 / NumPy 2, its mdct/pacfileThem/bitpack/huffman are SyntaxErrors (py2 `print`) and the rest raises on
float sizes.  Every place where Python-2 semantics change a result (integer `/`, float used as a size, int()
truncation) is restated explicitly and marked `py2:` in the code.

PARITY PINNING STATUS (round 2: every stage is pinned by outputs of the reference's own code)
    tests/golden/make_golden.py -- plain import of the py3-importable modules: quantize.*, bitalloc.BitAlloc,
        ms_stereo.*, window.HanningWindow  (quantize/bitalloc/ms_stereo/window/decode.npz).
    tests/golden/make_golden_ref.py -- the reference's function bodies executed through
        tests/golden/py2harness.py (one AST pass gives `/`, float sizes / indices, `dict.has_key` their Python 2 /
        NumPy < 1.12 meaning; nothing else is changed): psychoac primitives and band tables, KBDWindow /
        TransitionWindow, MDCT / IMDCT, getMaskedThreshold / CalcSMRs, EncodeSingleChannel / JointEncodeChannels /
        Encode / EncodeNoHuff / JointEncode chains with block switching, reservoir and Huffman gain, Decode /
        JointDecode  (ref_psychoac/window/mdct/smr/encode.npz).
    tests/golden/make_golden_pac.py -- pacfileThem.py executed AS A SCRIPT on synthetic WAV files: the .pac bytes
        its encode direction wrote (with and without Huffman tables) and the WAV its decode direction wrote
        (ref_pac.npz): pins WAV ingest, transient detector + look-ahead sequencing, chunk layouts, bit packing,
        header, Close() flush, chunk reader.
    tests/test_reference_golden.py holds the oracle to all of it: bit-exact, floats included (the oracle performs
    the same NumPy operations in the same order).
    Also: MDCT == MDCTslow (mdct.py:185-199), TDAC known-answer vector (mdct.py:131-182), bit-packer vector
    (bitpack.py:183-196).
    What stays unpinned: the last-ulp behaviour of the NumPy / libm build the authors ran (FFTPACK vs pocketfft,
    log10/arctan/power): the fixtures are this container's NumPy executing the reference's code.  The Huffman
    TABLES are re-derived from the text of the reference's pickles (tools/check_huffman_tables.py), and the files
    the harness hands to the reference's pickle.load are written from that data (the reference's own pickles are
    never unpickled).

Two flavours with identical results:
    oracle.<module>   "faithful": one block at a time, same redundancy as the reference (per-call
                      window construction, duplicate masked-threshold evaluation, Python peak
                      loop).  This is what bench.py times as the CPU baseline (kind="port").
    oracle.fast       batched / vectorised; same arithmetic, same summation orders; used by the
                      parity tests so 10^3..10^4 frames finish in seconds.
"""

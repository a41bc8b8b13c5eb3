"""
CPU-side checks of the boundary: the shared library loads, exports every function include/mrc_hip.h
declares, refuses to run without a GPU (no CPU fallback), and the host-side logic (Huffman stage,
synthetic inputs, table data) agrees with the oracle.  No kernel is launched here.
"""
import os
import re
import types

import numpy as np
import pytest

from oracle import codec as ocodec, huffman_tables as otables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "mrc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from mrcaudiocodec_amd import _lib
    import ctypes
    names = _header_functions()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libmrc_hip.so does not export %s" % n
        assert n in _lib.EXPORTS, "ctypes binding does not declare %s" % n
    assert _lib.lib.mrc_version() == 300


def test_no_cpu_fallback_without_gpu():
    from mrcaudiocodec_amd import Handle, MrcError, _lib
    if _lib.lib.mrc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(MrcError):
        Handle()
    import mrcaudiocodec_amd.codecThem as codec
    cp = ocodec.default_params()
    with pytest.raises(MrcError):
        codec.EncodeSingleChannel(np.zeros(2048), cp)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mrcaudiocodec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src, f


def test_huffman_tables_match_oracle():
    from mrcaudiocodec_amd import huffman_tables as pt
    assert pt.TABLE_NAMES == otables.TABLE_ORDER and pt.RAW_TABLE_ID == otables.RAW_TABLE_ID
    for name in pt.TABLE_NAMES:
        table, esc = otables.TABLES[name]
        assert pt.ESCAPE[name] == esc
        assert {v: c for v, (c, _) in table.items()} == pt.CODES[name]


def test_huffman_gain_host_logic_matches_oracle():
    import mrcaudiocodec_amd.codecThem as codec
    rng = np.random.default_rng(7)
    cp = ocodec.default_params()
    for trial in range(40):
        ba = rng.choice([0, 0, 2, 3, 4, 5, 7], size=25)
        n = int(np.sum(cp.sfBands.nLines[ba > 0]))
        scale = [1, 2, 4, 20][trial % 4]
        m = np.abs(np.rint(rng.laplace(0, scale, n))).astype(np.int32)
        m = np.minimum(m, 2 ** 15)
        got = codec.calculateHuffmanGain(m, ba, cp)
        want = ocodec.calculateHuffmanGain(m, ba, cp)
        assert got[0] == want[0] and got[2] == want[2]
        assert list(got[1]) == list(want[1])


def test_synth_shapes():
    from mrcaudiocodec_amd import synth
    assert synth.c2_noise(3).shape == (4 * 1024,) and not synth.c2_noise(3)[:1024].any()
    assert synth.c3_stereo(3).shape == (2, 4 * 1024)
    x, shapes = synth.c4_transients(10)
    assert shapes[4] == (4096, 1024, 128) and shapes[5][1:] == (128, 128) and shapes[12][1:] == (128, 1024)
    assert shapes[-1][0] + shapes[-1][1] + shapes[-1][2] == len(x)
    assert np.abs(x).max() <= 1.0
    assert synth.pcm_to_float([1])[0] == 3.0518043793392844e-05


def test_table_log10_accuracy(tmp_path):
    # mrc_log10.hpp (the SPL conversions of smr_kernel) compiled for the HOST: <= 0.6 ulp against long double
    # wherever |log10 x| >= 1/4 and <= 4e-17 absolute near x = 1 (the value is added to 96 afterwards)
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "log10_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "mrcaudiocodec_amd", "csrc"),
                           os.path.join(root, "tests", "log10_check.cpp"), "-o", exe])
    max_ulp, max_abs = map(float, subprocess.check_output([exe]).split())
    assert max_ulp <= 0.6, max_ulp
    assert max_abs <= 4e-17, max_abs


def test_pcm16_map_is_exact():
    """dev::pcm16_to_frac (csrc/mrc_device.hpp) replaces the reference's IEEE division 2c/65535 (pcmfile.py:91-100 via
    quantize.py:90-111) by q = fma(c, kHi, c kLo) with kHi + kLo = 2/65535 to 106 bits.  Checked here for EVERY 16-bit code
    in exact rational arithmetic (each device operation rounds once: Fraction -> float is that rounding), against the
    values the reference's own function returned for the recorded codes."""
    from fractions import Fraction
    k_hi, k_lo = float.fromhex("0x1.0001000100010p-15"), float.fromhex("0x1.0001000100010p-79")
    assert k_hi == float(Fraction(2, 65535)) and k_lo == float(Fraction(2, 65535) - Fraction(k_hi))

    def dev_map(c):
        n = 0 if c == -32768 else c
        low = float(Fraction(n) * Fraction(k_lo))
        return float(Fraction(n) * Fraction(k_hi) + Fraction(low))
    for c in range(-32768, 32768):
        want = 0.0 if c == -32768 else (2.0 * c) / 65535.0
        assert dev_map(c) == want, c
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_encode.npz"), allow_pickle=False)
    assert [dev_map(int(c)) for c in g["pcmmap_in"]] == list(g["pcmmap_out"])

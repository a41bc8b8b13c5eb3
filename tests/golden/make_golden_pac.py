#!/usr/bin/env python3
"""
File-level golden data from the reference's OWN command-line driver: pacfileThem.py is executed AS A SCRIPT
(`python pacfileThem.py in.wav`, pacfileThem.py:1064-1231: WAV ingest, transient detector with one block of
look-ahead, JointWriteDataBlock per block, Close() flush, then the decode direction) through
tests/golden/py2harness.py in the build container, on synthetic 16-bit stereo WAV files.  Recorded, as data:

    tests/golden/ref_pac.npz   <case>_pcm       int16 [2][n]   the WAV's samples
                               <case>_rate      sample rate
                               <case>_pac       uint8 []       the .pac file the reference wrote, Huffman tables present
                               <case>_pac_raw   uint8 []       the same without ./training_data (every block raw, id 15)
                               <case>_decoded   int16 [2][m]   the WAV its decode direction wrote

The Huffman files under ./training_data are written by the harness from this repo's table data (the reference's
pickles are never loaded), in the repo's fixed table order.
Known property of the reference's driver, visible in the data: the first 1024 decoded samples per channel are not
decoded audio but the LAST input block -- its decode loop writes the look-ahead buffer `dataMem` left over from
the encode direction (pacfileThem.py:1185-1189,1203-1214) -- so decoder tests compare from sample 1024 on.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

import py2harness as H                          # noqa: E402
from oracle import huffman_tables as HT          # noqa: E402   (table data)
from oracle import transient as otr, codec as ocodec, pcmfile as opcm   # noqa: E402   (only to ASSERT the cases contain short blocks)


def wav_bytes(pcm, rate):
    import struct
    data = np.ascontiguousarray(pcm.T).astype("<i2").tobytes()
    nch = pcm.shape[0]
    return (b"RIFF" + struct.pack("<L", 36 + len(data)) + b"WAVE" + b"fmt " +
            struct.pack("<LHHLLHH", 16, 1, nch, rate, rate * nch * 2, nch * 2, 16) + b"data" +
            struct.pack("<L", len(data)) + data)


def content(seed, n, rate, burst_at):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    g1, g2 = rng.normal(0, 0.1 * 32767, n), rng.normal(0, 0.1 * 32767, n)
    hop = t // 1024
    lvl = 10.0 ** (-1.5 * (hop % 5 == 3)) * 10.0 ** (-2.2 * (hop % 7 == 5))
    tone = np.sin(2 * np.pi * 440.0 / rate * t) * (hop > 9)
    left = g1 * lvl + 3000 * tone
    right = np.where(hop % 2 == 0, 0.8 * g1 + 0.2 * g2, 0.1 * g2) * lvl + 2000 * tone
    pcm = np.clip(np.rint(np.stack([left, right])), -32767, 32767).astype(np.int16)
    for p in burst_at:
        pcm[:, p:p + 128] = np.clip(rng.normal(0, 0.5 * 32767, (2, 128)), -32767, 32767).astype(np.int16)
    return pcm


CASES = {
    # ragged length (last block zero padded), one burst -> long / transition / 8 short / transition
    "a48": (48000, content(5, 14 * 1024 - 300, 48000, [6200])),
    # length an exact multiple of the block size (the header's inverted padding test, pacfileThem.py:595-597),
    # 44.1 kHz band tables, bursts in adjacent blocks, full-scale negative code -32768
    "b44": (44100, content(6, 12 * 1024, 44100, [3100, 4500])),
}
CASES["b44"][1][0, 100] = -32768
CASES["b44"][1][1, 7000] = -32768

out = {}
cwd = os.getcwd()
for name, (rate, pcm) in CASES.items():
    for with_tables in (True, False):
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                with open("in.wav", "wb") as f:
                    f.write(wav_bytes(pcm, rate))
                H.load_file_layer(REF)
                if with_tables:
                    H.write_huffman_files("./training_data/", HT.TABLES, HT.TABLE_ORDER)
                err = H.run_pacfile_main("in.wav", REF)
                if err is not None:
                    raise err
                pac = np.frombuffer(open("in.pac", "rb").read(), dtype=np.uint8)
                w = open("in_decoded.wav", "rb").read()
                dec = np.frombuffer(w[44:], dtype="<i2").reshape(-1, 2).T.astype(np.int16)
                if with_tables:
                    sr, nch, ns, hops = opcm.read_wav("in.wav")
                    cp = ocodec.default_params(sampleRate=sr, nChannels=2)
                    shapes = otr.block_shapes(np.concatenate([np.zeros((2, 1024)), hops], axis=1), cp)
                    assert any(b == 128 for (_o, _a, b) in shapes), "case %s has no short blocks" % name
            finally:
                os.chdir(cwd)
        out[name + ("_pac" if with_tables else "_pac_raw")] = pac
        if with_tables:
            out[name + "_decoded"] = dec
    out[name + "_pcm"], out[name + "_rate"] = pcm, np.array(rate)
    assert not np.array_equal(out[name + "_pac"], out[name + "_pac_raw"]), "no Huffman-coded block in case " + name
out["cases"] = np.array(sorted(CASES))
np.savez_compressed(os.path.join(HERE, "ref_pac.npz"), **out)
print("ref_pac.npz", os.path.getsize(os.path.join(HERE, "ref_pac.npz")) // 1024, "KiB",
      {k: v.shape for k, v in out.items() if k != "cases"})

"""
oracle.pcmfile -- 16-bit PCM WAV ingest as the reference's encoder sees it (TEST ORACLE).

Restates pcmfile.py:34-66 (ReadFileHeader: RIFF/WAVE, first 'fmt ' chunk, PCM, 16 bit, first 'data' chunk)
and pcmfile.py:68-102 (ReadDataBlock: blocks of nSamplesPerBlock samples per channel, last block zero padded,
int16 code c -> sign(c) * 2|c| / 65535 via quantize.vDequantizeUniform, so that -32768 -> 0.0).
"""
from struct import unpack

import numpy as np

from .quantize import vDequantizeUniform


def read_wav(path, nSamplesPerBlock=1024):
    """-> (sampleRate, nChannels, numSamples, hops float64 [nChannels][nHops*nSamplesPerBlock])."""
    with open(path, "rb") as fp:
        tag = fp.read(12)
        if tag[0:4] != b"RIFF" or tag[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        while True:
            tag = fp.read(4)
            if len(tag) < 4:
                raise ValueError("no 'fmt ' chunk")
            if tag == b"fmt ":
                break
        (formatSize, formatTag, nChannels, sampleRate, bytesPerSec, blockAlign, bitsPerSample) = \
            unpack("<LHHLLHH", fp.read(20))
        if formatTag != 1 or bitsPerSample != 16:
            raise ValueError("only 16-bit PCM WAV files are supported")
        while True:
            tag = fp.read(4)
            if len(tag) < 4:
                raise ValueError("no 'data' chunk")
            if tag == b"data":
                break
        numSamples = unpack('<L', fp.read(4))[0] // (nChannels * 2)
        raw = fp.read(numSamples * nChannels * 2)
    codes = np.frombuffer(raw, dtype="<i2").astype(np.int64)
    codes = codes[:(len(codes) // nChannels) * nChannels].reshape(-1, nChannels).T
    nHops = -(-codes.shape[1] // nSamplesPerBlock)
    padded = np.zeros((nChannels, nHops * nSamplesPerBlock), dtype=np.int64)
    padded[:, :codes.shape[1]] = codes
    neg = padded < 0
    x = vDequantizeUniform(np.abs(padded).astype(np.float64), 16)
    x[neg] *= -1.
    return sampleRate, nChannels, numSamples, x

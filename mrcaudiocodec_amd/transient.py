"""
Transient detector and block-shape sequencing of the reference's encoder CLI, for whole streams at once.

    pacfileThem.py:1146-1154   filter design (SciPy, as in the reference) and thresholds  -> design_sos, THRESHOLDS
    pacfileThem.py:1025-1056   TransientDetector: high-pass each hop from a zero state, peak per 128-sample
                               sub-block, ratio test                                      -> peaks on the GPU
                                                                                             (mrc_transient_peaks),
                                                                                             tests below
    pacfileThem.py:1182-1214   one-hop look-ahead: hop i is coded as 8 short blocks if sum(blksw_i) > 1 or
                               any(blksw_{i+1} == 1), else as one long block           -> block_shapes

The filtering (the numeric part: hops x channels x 1024 samples x 10 biquads) runs on the GPU, one thread
per (hop, channel); the threshold tests and the sequencing are O(hops) host logic in NumPy.
"""
import numpy as np

THRESHOLDS = (0.1, 0.075)                        # pacfileThem.py:1154


def design_sos(sample_rate):
    """pacfileThem.py:1146-1147 (same SciPy calls as the reference)."""
    from scipy import signal
    b, a = signal.cheby2(20, 40, 9000. / sample_rate, 'high')
    return signal.tf2sos(b, a)


def transient_positions(peaks, T=THRESHOLDS):
    """peaks [nHops][nCh][nSub+1] (mrc_transient_peaks) -> bool [nHops][nSub]: position i+1 is flagged in hop h
    (pacfileThem.py:1046-1056; P[ch][0] of a hop is the last sub-block peak of the previous hop, 0 at the start)."""
    peaks = np.asarray(peaks, dtype=np.float64)
    nHops, nCh, n1 = peaks.shape
    nSub = n1 - 1
    P = np.zeros((nHops, nCh, nSub + 1))
    P[:, :, 1:] = peaks[:, :, :nSub]
    P[1:, :, 0] = peaks[:-1, :, nSub - 1]
    loud = peaks[:, :, nSub] > T[0]
    jump = P[:, :, 1:] * T[1] > P[:, :, :-1]
    return np.any(jump & loud[:, :, None], axis=1)


def block_shapes(handle, stream, sos=None, T=THRESHOLDS):
    """(offset, a, b) of every block the reference's encode loop writes for `stream`
    [nCh][(nHops+1)*hop] (starting with the zero prior hop).  The last hop is analysed but never written
    (the reference's loop ends before it; Close() only flushes zeros)."""
    hop, n_short = handle.cfg.n_mdct_lines, handle.cfg.n_short
    sos = design_sos(handle.cfg.sample_rate) if sos is None else sos
    return shapes_from_flags(transient_positions(handle.transient_peaks(stream, sos), T), hop, n_short)


def block_shape_array(handle, stream, sos=None, T=THRESHOLDS):
    """block_shapes as ONE int64 array [nBlocks][3] = (offset, a, b); `stream` may hold the file's int16 PCM codes."""
    hop, n_short = handle.cfg.n_mdct_lines, handle.cfg.n_short
    sos = design_sos(handle.cfg.sample_rate) if sos is None else sos
    return shape_array_from_flags(transient_positions(handle.transient_peaks(stream, sos), T), hop, n_short)


def shapes_from_flags(flags, hop, n_short):
    """pacfileThem.py:1182-1214 given the per-hop transient positions (bool [nHops][nSub]) -> list of (offset, a, b)."""
    return [tuple(r) for r in shape_array_from_flags(flags, hop, n_short).tolist()]


def shape_array_from_flags(flags, hop, n_short):
    """The same as ONE int64 array [nBlocks][3] = (offset, a, b) per block, without a Python loop over the hops: hop i
    (the last one excepted: it is analysed but never written) becomes nSub short blocks if sum(blksw_i) > 1 or
    any(blksw_{i+1} == 1), else one long block; `a` of a block is the `b` of the block before it, the offset the sum of
    the a's before it (pacfileThem.py:1192-1210)."""
    flags = np.asarray(flags, dtype=bool)
    nSub = hop // n_short
    if flags.shape[0] < 2:
        return np.zeros((0, 3), np.int64)
    pos = np.arange(1, nSub + 1)
    sum_pos = (flags * pos).sum(axis=1)
    short = (sum_pos[:-1] > 1) | flags[1:, 0]                          # per written hop
    b = np.where(np.repeat(short, np.where(short, nSub, 1)), n_short, hop).astype(np.int64)
    a = np.empty_like(b)
    a[0] = hop
    a[1:] = b[:-1]
    off = np.zeros_like(b)
    off[1:] = np.cumsum(a)[:-1]
    return np.stack([off, a, b], axis=1)


def block_shape_array_dev(handle, streams_ptr, sample_format, n_hops, n_channels, channel_stride, sos=None, T=THRESHOLDS,
                          stream=None):
    """block_shapes for a stream that is already in DEVICE memory (int16 PCM codes or float64): the filtering runs there
    (mrc_dev_transient_peaks), only the peaks (9 doubles per hop and channel) come back.  -> int64 [nBlocks][3]."""
    import torch
    hop, n_short = handle.cfg.n_mdct_lines, handle.cfg.n_short
    sos = design_sos(handle.cfg.sample_rate) if sos is None else sos
    dev = torch.device("cuda", handle.cfg.device_id)
    peaks = torch.empty((n_hops, n_channels, hop // n_short + 1), dtype=torch.float64, device=dev)
    handle.dev_transient_peaks(n_hops, n_channels, sos, streams_ptr, sample_format, channel_stride, peaks.data_ptr(),
                               torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
    return shape_array_from_flags(transient_positions(peaks.cpu().numpy(), T), hop, n_short)

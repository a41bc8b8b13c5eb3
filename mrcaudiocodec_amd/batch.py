"""
Device-resident batch / stream encoding on top of the C ABI's `mrc_dev_*` entry points.  PyTorch is
used only as plumbing: HBM allocations (torch tensors), the current HIP stream, and -- in bench.py --
torch.distributed for the barrier.  All computation happens in libmrc_hip.so.

Stream layout (what pacfileThem.py:628-631 does one block at a time): a channel is one contiguous
float64 array that starts with the prior hop (zeros at file start); long frame f covers samples
[f*1024, f*1024 + 2048), so every hop is stored once and frames overlap by 50 %.
Frames are independent given `reservoir_in` (zeros unless supplied) -- SURVEY.md 8(e): a batch shards
across GPUs as contiguous frame ranges with a one-hop halo and no collective.
"""
import torch

from ._lib import Handle


def _ptr(t):
    return None if t is None else t.data_ptr()


class StreamEncoder:
    def __init__(self, handle=None, device_id=0, **codec_params):
        self.h = handle if handle is not None else Handle(device_id=device_id, **codec_params)
        self.device = torch.device("cuda", self.h.cfg.device_id)
        self._out = {}

    def _alloc(self, n, joint, nb, half, mant16=False):
        nsig, nstream = (4, 2) if joint else (1, 1)
        i32 = dict(dtype=torch.int32, device=self.device)
        return dict(overall_scale=torch.empty((n, nsig), **i32),
                    ms_switch=torch.empty((n, nb), **i32) if joint else None,
                    bit_alloc=torch.empty((n, nstream, nb), **i32),
                    scale_factor=torch.empty((n, nstream, nb), **i32),
                    # uint16 codes travel as int16 storage (torch has no general uint16): view as uint16 on the host
                    mantissa=torch.empty((n, nstream, half), dtype=torch.int16 if mant16 else torch.int32, device=self.device),
                    reservoir_out=torch.empty((n,), **i32))

    def _outputs(self, n, joint, nb, half, mant16=False):
        key = (n, joint, nb, half, mant16)
        if key not in self._out:
            if len(self._out) >= 8:                           # a block-switched stream cycles through four shapes
                self._out.clear()
            self._out[key] = self._alloc(n, joint, nb, half, mant16)
        return self._out[key]

    def encode(self, a, b, left, right, n_frames, frame_stride, offsets=None, reservoir_in=None, lines_out=None,
               fresh=False, offsets_checked=False, mantissa16=False):
        """Encode n_frames blocks of shape (a,b) read from device tensor(s) `left` (and `right` for joint stereo):
        float64 signed fractions or int16 PCM codes (converted on load as pcmfile.py:91-100 does).  mantissa16: the
        mantissa plane as 16-bit codes (int16 storage of uint16 values).  Returns a dict of device tensors (reused
        between calls of the same size unless fresh=True)."""
        for t in (left, right, offsets, reservoir_in, lines_out):
            if t is not None and (not t.is_cuda or not t.is_contiguous()):
                raise ValueError("device-contiguous tensors expected")
        if left.dtype not in (torch.float64, torch.int16) or (right is not None and right.dtype != left.dtype):
            raise ValueError("PCM must be float64 signed fractions or int16 codes (both channels alike)")
        if not (offsets is not None and offsets_checked):    # (reading offsets back synchronises with the device)
            last = (offsets.max().item() if offsets is not None else (n_frames - 1) * frame_stride) + a + b
            if n_frames > 0 and (left.numel() < last or (right is not None and right.numel() < last)):
                raise ValueError("stream too short for %d frames" % n_frames)
        nb = len(self.h.bands(a, b))
        out = (self._alloc if fresh else self._outputs)(n_frames, right is not None, nb, (a + b) // 2, mantissa16)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.h.dev_encode_ex(a, b, n_frames, _ptr(left), _ptr(right), 1 if left.dtype == torch.int16 else 0, frame_stride,
                             _ptr(offsets), _ptr(reservoir_in), _ptr(out["overall_scale"]), _ptr(out["ms_switch"]),
                             _ptr(out["bit_alloc"]), _ptr(out["scale_factor"]), _ptr(out["mantissa"]),
                             1 if mantissa16 else 0, _ptr(out["reservoir_out"]), _ptr(lines_out), stream)
        return {k: v for k, v in out.items() if v is not None}

    def encode_long(self, left, right, n_frames, reservoir_in=None, mantissa16=False):
        """All-long-block stream (a = b = nMDCTLines), hop-overlapped layout."""
        L = self.h.cfg.n_mdct_lines
        return self.encode(L, L, left, right, n_frames, L, None, reservoir_in, mantissa16=mantissa16)

    def pack(self, a, b, out, use_huffman=True, huff_table=None, typical_bytes=None):
        """`.pac` chunks of the blocks in `out` (a dict from encode), packed ON THE DEVICE (mrc_dev_pack_blocks): the bytes
        pacfileThem.py's WriteDataBlock / JointWriteDataBlock append, block after block.  huff_table: device tensor of
        table ids already chosen (huffman_gain), else priced here (or raw).  -> dict of device tensors: bytes (uint8, the
        used prefix of the buffer), block_offset [n + 1] int64, huff_table, bits_saved [n][nStreams]."""
        from . import _lib
        n, ns = out["bit_alloc"].shape[0], out["bit_alloc"].shape[1]
        joint = out["overall_scale"].shape[-1] == 4 and ns == 2
        m16 = out["mantissa"].dtype == torch.int16
        cfg = self.h.cfg
        bound = int(_lib.lib.mrc_pack_bound(_lib.C.byref(cfg), int(a), int(b), 1, 1 if joint else 0))
        i32 = dict(dtype=torch.int32, device=self.device)
        table, saved = torch.empty((n, ns), **i32), torch.empty((n, ns), **i32)
        offs = torch.empty((n + 1,), dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # a first buffer sized for the typical chunk (a block of 1024 lines at 128 kb/s/ch is ~350 bytes per channel), the
        # worst case (every line 16 raw bits behind the longest escape code) only if that turns out too small
        cap = n * ns * (typical_bytes or max(64, (a + b) // 2)) + 4096
        for attempt in (0, 1):
            buf = torch.empty((cap,), dtype=torch.uint8, device=self.device)
            try:
                total = self.h.dev_pack_blocks(a, b, n, ns, joint, use_huffman, _ptr(huff_table), _ptr(out["overall_scale"]),
                                               _ptr(out["ms_switch"]) if joint else None, _ptr(out["scale_factor"]),
                                               _ptr(out["bit_alloc"]), _ptr(out["mantissa"]), m16, _ptr(buf), cap, _ptr(offs),
                                               _ptr(table), _ptr(saved), True, stream)
                break
            except _lib.MrcError as e:
                # only "buffer too small" is worth a second attempt at the worst-case size; a bad table id or a HIP error is not
                if attempt or cap >= n * ns * bound or getattr(e, "code", None) != _lib.MRC_ERR_NOMEM:
                    raise
                cap = n * ns * bound
        # with the tables GIVEN nothing is priced here: no bits_saved (mirrors pacfile._pack)
        return {"bytes": buf[:total], "block_offset": offs, "huff_table": table,
                "bits_saved": None if huff_table is not None else saved}

    def huffman_gain(self, a, b, out, use_huffman=True):
        """Prices the Huffman tables for the blocks in `out` (a dict from encode) on the device and returns
        (huff_table [n][nStreams], bits_saved [n][nStreams], reservoir_next [n]) -- codecThem.py:136-203,224,274."""
        n, ns = out["bit_alloc"].shape[0], out["bit_alloc"].shape[1]
        i32 = dict(dtype=torch.int32, device=self.device)
        if not use_huffman:                                     # EncodeNoHuff: raw mantissas, nothing saved
            return (torch.full((n, ns), 15, **i32), torch.zeros((n, ns), **i32), out["reservoir_out"].clone())
        table, saved, nxt = torch.empty((n, ns), **i32), torch.empty((n, ns), **i32), torch.empty((n,), **i32)
        self.h.dev_huffman_gain(a, b, n, ns, _ptr(out["bit_alloc"]), _ptr(out["mantissa"]), _ptr(out["reservoir_out"]),
                                _ptr(table), _ptr(saved), _ptr(nxt), torch.cuda.current_stream(self.device).cuda_stream)
        return table, saved, nxt

    def encode_chained_pac(self, left, right, shapes, use_huffman=True, with_flush=True, num_samples=None, reservoir_in=None,
                           out=None, want_items=False):
        """The reference's whole encode loop for stereo streams that are RESIDENT in HBM (mrc_dev_encode_chained_pac): left /
        right [nStreams][stride] device tensors (int16 PCM codes or float64), shapes[s] = the (offset, a, b) sequence of
        stream s (lists or int arrays [n][3]).  -> dict: bytes (device uint8 tensor, the used prefix), stream_offset /
        reservoir_out (host arrays), item_offset (with want_items, else None), total; `out`: a device uint8 tensor to write
        into (else allocated at the worst-case size)."""
        from . import _lib
        if left.dim() == 1:
            left, right = left[None], right[None]
        for t in (left, right):
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("device-contiguous tensors expected")
        if left.dtype not in (torch.float64, torch.int16) or right.dtype != left.dtype or right.shape != left.shape:
            raise ValueError("left / right: float64 or int16 [nStreams][stride], alike")
        if not isinstance(shapes, _lib.ChainSchedule):
            shapes = _lib.ChainSchedule(shapes)                          # (build it once yourself if you call repeatedly)
        if len(shapes) != left.shape[0]:
            raise ValueError("one shape list per stream expected")
        if out is None:
            start, _, a, b = self.h._chain_schedule(shapes)
            C = _lib.C
            bound = int(_lib.lib.mrc_chain_out_bound(self.h._h, len(shapes), start.ctypes.data_as(_lib._i64p),
                                                     a.ctypes.data_as(_lib._i32p), b.ctypes.data_as(_lib._i32p),
                                                     1 if with_flush else 0, 0 if num_samples is None else 1))
            if bound < 0:
                raise _lib.MrcError("mrc_chain_out_bound failed (%d)" % bound)
            out = torch.empty((max(bound, 1),), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        r = self.h.encode_chained_pac(None, None, shapes, use_huffman=use_huffman, with_flush=with_flush,
                                      num_samples=num_samples, reservoir_in=reservoir_in,
                                      device=(left.data_ptr(), right.data_ptr(), 1 if left.dtype == torch.int16 else 0,
                                              left.shape[1], out.data_ptr(), out.numel()), stream=stream,
                                      want_items=want_items)
        r["bytes"] = out[:r["total"]]
        return r

    def encode_chained(self, left, right, shapes, use_huffman=True):
        """Stream mode for MANY stereo streams at once: left/right [nStreams][samples] on the device (each row
        starts with its zero prior hop), shapes[s] = the (offset, a, b) sequence of stream s (lists, or ONE int64 array [nStreams][nBlocks][3]).  Step t encodes the
        t-th block of every stream that still has one, grouped by block shape; the bit reservoir of each stream
        is chained from block to block ON THE DEVICE (reservoir_out + Huffman bits_saved, codecThem.py:274,503),
        so there is no host round trip inside the loop.  Returns (steps, reservoir): steps = list of
        (stream ids (int64 array), a, b, outputs dict of device tensors) in encode order; reservoir [nStreams] int32."""
        import numpy as np
        nS, stride = left.shape[0], left.shape[1]
        if right.shape != left.shape or len(shapes) != nS:
            raise ValueError("left/right [nStreams][samples] and one shape list per stream expected")
        flatL, flatR = left.reshape(-1), right.reshape(-1)
        reservoir = torch.zeros((nS,), dtype=torch.int32, device=self.device)
        # the schedule (which streams take which block shape at which step) is host logic on the shape lists: built
        # with NumPy and uploaded BEFORE the loop, so the loop itself only queues kernels
        tab = None
        if isinstance(shapes, np.ndarray) and shapes.ndim == 3 and shapes.shape[2] == 3:
            tab = np.ascontiguousarray(shapes, dtype=np.int64)   # the schedule as an array [nStreams][nBlocks][3]: no Python loop
            nT = tab.shape[1]                                    # (rows of (-1, -1, -1) pad shorter streams)
        else:
            nT = max((len(s) for s in shapes), default=0)
        if tab is None and nT and all(len(s) == nT for s in shapes):   # equally long streams: one conversion
            try:
                tab = np.asarray(shapes, dtype=np.int64)
                if tab.shape != (nS, nT, 3):
                    tab = None
            except (ValueError, TypeError):
                tab = None
        if tab is None:
            tab = np.full((nS, nT, 3), -1, dtype=np.int64)
            for s, sh in enumerate(shapes):
                if len(sh):
                    tab[s, :len(sh)] = np.asarray(sh, dtype=np.int64)
        if nT and (tab[:, :, 0] + tab[:, :, 1] + tab[:, :, 2] > stride).any():
            raise ValueError("a stream is too short for one of its blocks")
        plan = []
        for t in range(nT):
            live = tab[:, t, 1] > 0
            code = tab[:, t, 1] * (1 << 32) + tab[:, t, 2]                    # (a, b) as one sortable key
            for key in np.unique(code[live]).tolist():
                a, b = key >> 32, key & 0xffffffff
                ids = np.nonzero(live & (code == key))[0]
                plan.append((int(a), int(b), ids, ids * stride + tab[ids, t, 0]))
        dev_plan = [(a, b, ids, torch.from_numpy(ids).to(self.device), torch.from_numpy(offs).to(self.device))
                    for (a, b, ids, offs) in plan]
        steps = []
        for (a, b, ids, idx, offs) in dev_plan:
            whole = len(ids) == nS                             # every stream takes this shape: no gather / scatter
            out = self.encode(a, b, flatL, flatR, len(ids), 0, offs, reservoir if whole else reservoir[idx].contiguous(),
                              fresh=True, offsets_checked=True)
            out["huff_table"], _, nxt = self.huffman_gain(a, b, out, use_huffman)    # kept: the packer need not price again
            if whole:
                reservoir = nxt
            else:
                reservoir[idx] = nxt
            steps.append((ids, a, b, out))
        return steps, reservoir

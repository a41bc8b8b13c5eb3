"""
ctypes binding of libmrc_hip.so (include/mrc_hip.h).  Thin: argument marshalling and error mapping
only.  There is no fallback of any kind: if the shared library is missing the import fails, and if no
gfx950 device is usable `Handle()` raises -- nothing in this package computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRC_HIP_LIBRARY") or os.path.join(_HERE, "libmrc_hip.so")   # override: profiling builds

MRC_MAX_BANDS = 32
MRC_ERR_NOMEM = -4
# array arguments travel as plain addresses (c_void_p prototypes): numpy's typed `data_as` costs ~2.3 us per array, which
# at thirteen arrays per call was a third of a one-block call through the drop-in seam; the names say what the C side expects
_i32p = _i64p = _f64p = _u8p = C.c_void_p


class MrcConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_mdct_lines", C.c_int32), ("n_short", C.c_int32),
                ("n_scale_bits", C.c_int32), ("n_mant_size_bits", C.c_int32), ("blksw_bits_a", C.c_int32),
                ("blksw_bits_b", C.c_int32), ("device_id", C.c_int32), ("target_bits_per_sample", C.c_double)]


class MrcError(RuntimeError):
    """code: the mrc_status the library returned (None for errors raised by the binding itself)"""
    code = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7); if libmrc_hip.so pulled in /opt/rocm's copy first, a later `import torch` would
    load a second runtime that cannot see the GPU, and device pointers / streams could not be shared.
    So when torch is installed, its copy is loaded first (by path, nothing of torch is imported) and
    satisfies libmrc_hip.so's NEEDED entry; otherwise the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mrcaudiocodec_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mrcaudiocodec_amd/csrc` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    sig = {
        "mrc_version": (C.c_int, []),
        "mrc_device_count": (C.c_int, []),
        "mrc_default_config": (None, [C.POINTER(MrcConfig)]),
        "mrc_create": (C.c_int, [C.POINTER(MrcConfig), C.POINTER(H)]),
        "mrc_destroy": (None, [H]),
        "mrc_last_error": (C.c_char_p, [H]),
        "mrc_shape_bands": (C.c_int, [H, C.c_int, C.c_int, _i32p, _i32p]),
        "mrc_shape_budget": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int32, _f64p]),
        "mrc_encode_mono": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, _i32p, _i32p, _i32p, _i32p, _i32p,
                                      _i32p, _f64p]),
        "mrc_encode_joint": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, _f64p, _i32p, _i32p, _i32p, _i32p,
                                       _i32p, _i32p, _i32p, _f64p]),
        "mrc_window": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, _f64p]),
        "mrc_mdct": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, C.c_int, _f64p, _i32p]),
        "mrc_smr": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, _f64p, _i32p, _f64p, _f64p]),
        "mrc_bitalloc": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _i32p, _f64p, _f64p, _i32p, _i32p]),
        "mrc_bitalloc_inplace": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _i32p, _f64p, _f64p, _i32p, _i32p]),
        "mrc_scale_factor": (C.c_int, [H, C.c_int64, C.c_int, _f64p, _i32p, _i32p]),
        "mrc_mantissa": (C.c_int, [H, C.c_int64, C.c_int, _f64p, _i32p, _i32p, _i32p]),
        "mrc_transient_peaks": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, _f64p, _f64p]),
        "mrc_transient_peaks_ex": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, C.c_void_p, C.c_int, _f64p]),
        "mrc_dev_transient_peaks": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, _f64p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p,
                                              C.c_void_p]),
        "mrc_stereo_masking_factor": (C.c_int, [H, C.c_int64, _f64p, _f64p, _f64p, _f64p, _f64p]),
        "mrc_ms_switch": (C.c_int, [H, C.c_int64, C.c_int, _i32p, _f64p, _f64p, _i32p]),
        "mrc_dev_mdct": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
        "mrc_dev_smr": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "mrc_dev_alloc_quant": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 10),
        "mrc_dev_encode": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64] +
                           [C.c_void_p] * 10),
        "mrc_dev_encode_ex": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int64] +
                              [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 3),
        "mrc_encode_mono_blocks": (C.c_int, [H, C.c_int64, _f64p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p]),
        "mrc_encode_joint_blocks": (C.c_int, [H, C.c_int64, _f64p, _f64p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p,
                                              _i32p, _i32p]),
        "mrc_encode_stream_pcm16": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
        "mrc_encode_stream_pcm16_pac": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64p, C.c_int64]),
        "mrc_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
        "mrc_host_free": (C.c_int, [C.c_void_p]),
        "mrc_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
        "mrc_host_unregister": (C.c_int, [C.c_void_p]),
        "mrc_pcm_to_float": (C.c_int, [H, C.c_int64, C.c_void_p, _f64p]),
        "mrc_quantize_uniform": (C.c_int, [H, C.c_int64, C.c_int, _f64p, _i64p]),
        "mrc_bark": (C.c_int, [H, C.c_int64, _f64p, _f64p]),
        "mrc_get_kernel_ms": (C.c_int, [H, _f64p]),
        "mrc_huffman_gain": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p]),
        "mrc_dev_huffman_gain": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 7),
        "mrc_dev_pack_blocks": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 +
                                [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, _i64p, C.c_void_p]),
        "mrc_dev_pack_status": (C.c_int, [H, _i64p, C.c_void_p]),
        "mrc_chain_out_bound": (C.c_int64, [H, C.c_int64, _i64p, _i32p, _i32p, C.c_int, C.c_int]),
        "mrc_encode_chained_stream_pcm16_pac": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, _i64p, _i64p, _i32p,
                                                          _i32p, _i32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                                          _i64p, _i64p, _i32p, _i32p, _i64p]),
        "mrc_encode_chained_stream_pac": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, _i64p, _i64p,
                                                    _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                                    _i64p, _i64p, _i32p, _i32p, _i64p]),
        "mrc_dev_encode_chained_pac": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, _i64p, _i64p,
                                                 _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                                 _i64p, _i64p, _i32p, _i32p, _i64p, C.c_void_p]),
        "mrc_get_chain_ms": (C.c_int, [H, _f64p]),
        "mrc_pac_read_header": (C.c_int, [_u8p, C.c_int64, C.POINTER(MrcConfig), _i32p, C.POINTER(C.c_uint32), _i64p]),
        "mrc_pac_scan_chunks": (C.c_int64, [_u8p, C.c_int64, C.c_int64, _i64p, C.c_int64]),
        "mrc_unpack_blocks": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, _u8p, C.c_int64, _i64p] +
                              [_i32p] * 8),
        "mrc_decode": (C.c_int, [H, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, _f64p]),
        "mrc_dev_decode": (C.c_int, [H, C.c_int, C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 9),
        "mrc_pcm16": (C.c_int, [H, C.c_int64, _f64p, C.POINTER(C.c_int16)]),
        "mrc_dev_pcm16": (C.c_int, [H, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
        "mrc_band_table": (C.c_int, [C.POINTER(MrcConfig), C.c_int, C.c_int, _i32p, _i32p]),
        "mrc_pack_bound": (C.c_int64, [C.POINTER(MrcConfig), C.c_int, C.c_int, C.c_int, C.c_int]),
        "mrc_pac_header": (C.c_int, [C.POINTER(MrcConfig), C.c_int, C.c_uint32, _u8p, C.c_int64, _i64p]),
        "mrc_pack_set_threads": (C.c_int, [C.c_int]),
        "mrc_pack_get_threads": (C.c_int, []),
        "mrc_pack_blocks_ex": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p,
                                         _i32p, _i32p, _i32p, _i32p, C.c_void_p, C.c_int, _u8p, C.c_int64, _i64p, _i32p, _i32p]),
        "mrc_pack_blocks_with_tables": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p,
                                                  _i32p, _i32p, _i32p, _u8p, C.c_int64, _i64p]),
        "mrc_pack_joint_blocks_with_tables": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, _i32p, _i32p,
                                                        _i32p, _i32p, _i32p, _i32p, _u8p, C.c_int64, _i64p]),
        "mrc_pack_blocks": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _i32p,
                                      _i32p, _i32p, _i32p, _u8p, C.c_int64, _i64p, _i32p, _i32p]),
        "mrc_pack_joint_blocks": (C.c_int, [C.POINTER(MrcConfig), C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p,
                                            _i32p, _i32p, _i32p, _u8p, C.c_int64, _i64p, _i32p, _i32p]),
        "mrc_chain_fetch_output": (C.c_int, [H, C.c_void_p, C.c_int64, _i64p]),
        "mrc_get_sensitivity": (C.c_int, [H, _i64p, C.c_int]),
        "mrc_set_timing": (C.c_int, [H, C.c_int]),
        "mrc_set_option": (C.c_int, [H, C.c_int, C.c_int]),
        "mrc_get_option": (C.c_int, [H, C.c_int, _i32p]),
        "mrc_get_stage_ms": (C.c_int, [H, _f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the library does not export the header's symbol
        fn.restype = res
        fn.argtypes = args
    return lib, tuple(sig)


lib, EXPORTS = _load()


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _reservoir(reservoir_in, n):
    """reservoir_in as int32 [n] (None stays None); the C side reads n values from it."""
    if reservoir_in is None:
        return None
    r = _i32(np.asarray(reservoir_in).reshape(-1))
    if r.shape != (n,):
        raise ValueError("reservoir_in must hold one value per block (%d), got %d" % (n, r.size))
    return r


def _p(a, typ):
    return None if a is None else a.ctypes.data


class ChainSchedule:
    """The block schedule of a chained encode as the C ABI takes it (block_start [nStreams + 1], offset / a / b per block),
    built ONCE from per-stream shape lists or arrays and reused from call to call."""

    def __init__(self, shapes):
        self.start, self.offset, self.a, self.b = Handle._chain_schedule(shapes)

    def __len__(self):
        return len(self.start) - 1


class Handle:
    """One mrc_handle: a device, a stream, the constant tables of the block shapes."""

    def __init__(self, sample_rate=48000, n_mdct_lines=1024, n_short=128, n_scale_bits=4, n_mant_size_bits=4,
                 target_bits_per_sample=2.86, blksw_bits_a=1, blksw_bits_b=1, device_id=0):
        cfg = MrcConfig()
        lib.mrc_default_config(C.byref(cfg))
        cfg.sample_rate = int(sample_rate)
        cfg.n_mdct_lines = int(n_mdct_lines)
        cfg.n_short = int(n_short)
        cfg.n_scale_bits = int(n_scale_bits)
        cfg.n_mant_size_bits = int(n_mant_size_bits)
        cfg.target_bits_per_sample = float(target_bits_per_sample)
        cfg.blksw_bits_a = int(blksw_bits_a)
        cfg.blksw_bits_b = int(blksw_bits_b)
        cfg.device_id = int(device_id)
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = lib.mrc_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = lib.mrc_last_error(None)
            self._h = None
            raise MrcError("mrc_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self._bands = {}

    def close(self):
        if getattr(self, "_h", None):
            lib.mrc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = lib.mrc_last_error(self._h)
            err = MrcError("libmrc_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
            err.code = rc
            raise err

    # ---- shape queries
    def bands(self, a, b):
        """nLines per scale-factor band of shape (a,b) (int32 array)."""
        key = (int(a), int(b))
        if key not in self._bands:
            n = C.c_int32()
            buf = np.zeros(MRC_MAX_BANDS, dtype=np.int32)
            self._check(lib.mrc_shape_bands(self._h, key[0], key[1], C.byref(n), _p(buf, _i32p)))
            self._bands[key] = buf[:n.value].copy()
        return self._bands[key]

    def budget(self, a, b, joint, reservoir=0):
        v = C.c_double()
        self._check(lib.mrc_shape_budget(self._h, int(a), int(b), int(bool(joint)), int(reservoir), C.byref(v)))
        return v.value

    # ---- host entry points
    def encode_mono(self, blocks, a, b, reservoir_in=None, want_mdct=False):
        blocks = _f64(blocks)
        n, N = blocks.shape
        if N != a + b:
            raise ValueError("blocks must be [n][a+b]")
        nb, half = len(self.bands(a, b)), N // 2
        res_in = _reservoir(reservoir_in, n)
        out = dict(overall_scale=np.empty(n, np.int32), scale_factor=np.empty((n, nb), np.int32),
                   bit_alloc=np.empty((n, nb), np.int32), mantissa=np.empty((n, half), np.int32),
                   reservoir_out=np.empty(n, np.int32))
        mdct = np.empty((n, half), np.float64) if want_mdct else None
        self._check(lib.mrc_encode_mono(self._h, n, a, b, _p(blocks, _f64p), _p(res_in, _i32p),
                                        _p(out["overall_scale"], _i32p), _p(out["scale_factor"], _i32p),
                                        _p(out["bit_alloc"], _i32p), _p(out["mantissa"], _i32p),
                                        _p(out["reservoir_out"], _i32p), _p(mdct, _f64p)))
        if want_mdct:
            out["mdct"] = mdct
        return out

    def encode_joint(self, left, right, a, b, reservoir_in=None, want_mdct=False):
        left, right = _f64(left), _f64(right)
        n, N = left.shape
        if N != a + b or right.shape != left.shape:
            raise ValueError("left/right must be [n][a+b]")
        nb, half = len(self.bands(a, b)), N // 2
        res_in = _reservoir(reservoir_in, n)
        out = dict(overall_scale=np.empty((n, 4), np.int32), ms_switch=np.empty((n, nb), np.int32),
                   scale_factor=np.empty((n, 2, nb), np.int32), bit_alloc=np.empty((n, 2, nb), np.int32),
                   mantissa=np.empty((n, 2, half), np.int32), reservoir_out=np.empty(n, np.int32))
        mdct = np.empty((n, 4, half), np.float64) if want_mdct else None
        self._check(lib.mrc_encode_joint(self._h, n, a, b, _p(left, _f64p), _p(right, _f64p), _p(res_in, _i32p),
                                         _p(out["overall_scale"], _i32p), _p(out["ms_switch"], _i32p),
                                         _p(out["scale_factor"], _i32p), _p(out["bit_alloc"], _i32p),
                                         _p(out["mantissa"], _i32p), _p(out["reservoir_out"], _i32p),
                                         _p(mdct, _f64p)))
        if want_mdct:
            out["mdct"] = mdct
        return out

    def encode_blocks(self, left, a, b, right=None, reservoir_in=None):
        """Blocks of MIXED shapes in one call (mrc_encode_mono_blocks / mrc_encode_joint_blocks): `left` (and `right`)
        is a list of 1-D arrays, block i holding a[i] + b[i] samples.  Outputs have fixed strides: scale_factor /
        bit_alloc [n][streams][MRC_MAX_BANDS], ms_switch [n][MRC_MAX_BANDS], mantissa [n][streams][n_mdct_lines]."""
        a, b = _i32(a), _i32(b)
        n = len(a)
        if len(b) != n or len(left) != n or (right is not None and len(right) != n):
            raise ValueError("one (a, b) pair and one block per entry expected")
        for i in range(n):
            if len(left[i]) != a[i] + b[i] or (right is not None and len(right[i]) != a[i] + b[i]):
                raise ValueError("block %d does not hold a + b = %d samples" % (i, a[i] + b[i]))
        packed_l = _f64(np.concatenate([np.asarray(x, dtype=np.float64) for x in left])) if n else np.zeros(0)
        res_in = _reservoir(reservoir_in, n)
        L = self.cfg.n_mdct_lines
        ns, nsig = (2, 4) if right is not None else (1, 1)
        out = dict(overall_scale=np.empty((n, nsig) if right is not None else n, np.int32),
                   scale_factor=np.empty((n, ns, MRC_MAX_BANDS), np.int32), bit_alloc=np.empty((n, ns, MRC_MAX_BANDS), np.int32),
                   mantissa=np.empty((n, ns, L), np.int32), reservoir_out=np.empty(n, np.int32))
        if right is None:
            self._check(lib.mrc_encode_mono_blocks(self._h, n, _p(packed_l, _f64p), _p(a, _i32p), _p(b, _i32p),
                                                   _p(res_in, _i32p), _p(out["overall_scale"], _i32p),
                                                   _p(out["scale_factor"], _i32p), _p(out["bit_alloc"], _i32p),
                                                   _p(out["mantissa"], _i32p), _p(out["reservoir_out"], _i32p)))
        else:
            packed_r = _f64(np.concatenate([np.asarray(x, dtype=np.float64) for x in right])) if n else np.zeros(0)
            out["ms_switch"] = np.empty((n, MRC_MAX_BANDS), np.int32)
            self._check(lib.mrc_encode_joint_blocks(self._h, n, _p(packed_l, _f64p), _p(packed_r, _f64p), _p(a, _i32p),
                                                    _p(b, _i32p), _p(res_in, _i32p), _p(out["overall_scale"], _i32p),
                                                    _p(out["ms_switch"], _i32p), _p(out["scale_factor"], _i32p),
                                                    _p(out["bit_alloc"], _i32p), _p(out["mantissa"], _i32p),
                                                    _p(out["reservoir_out"], _i32p)))
        return out

    def encode_stream_pcm16(self, pcm_left, pcm_right=None, reservoir_in=None, chunk_frames=0, out=None):
        """mrc_encode_stream_pcm16: int16 PCM stream(s) [(n+1) * L] in host memory -> codes in host memory, long
        blocks, pipelined over three HIP streams.  `out` may hold preallocated (ideally page-locked, see
        pinned_empty) arrays to write into; the mantissa plane is uint16."""
        L = self.cfg.n_mdct_lines
        pl = np.ascontiguousarray(pcm_left, dtype=np.int16)
        n = pl.size // L - 1
        if pl.ndim != 1 or pl.size != (n + 1) * L or n < 0:
            raise ValueError("pcm_left must be int16 [(n_frames + 1) * %d]" % L)
        joint = pcm_right is not None
        pr = None
        if joint:
            pr = np.ascontiguousarray(pcm_right, dtype=np.int16)
            if pr.shape != pl.shape:
                raise ValueError("pcm_right must match pcm_left")
        nb = len(self.bands(L, L))
        ns, nsig = (2, 4) if joint else (1, 1)
        want = dict(overall_scale=((n, nsig), np.int32), scale_factor=((n, ns, nb), np.int32),
                    bit_alloc=((n, ns, nb), np.int32), mantissa=((n, ns, L), np.uint16), reservoir_out=((n,), np.int32))
        if joint:
            want["ms_switch"] = ((n, nb), np.int32)
        res = {}
        for k, (shape, dt) in want.items():
            arr = None if out is None else out.get(k)
            if arr is None:
                arr = np.empty(shape, dt)
            elif arr.shape != shape or arr.dtype != dt or not arr.flags.c_contiguous:
                raise ValueError("out[%r] must be a C-contiguous %s array of shape %s" % (k, np.dtype(dt).name, shape))
            res[k] = arr
        res_in = _reservoir(reservoir_in, n)
        vp = lambda arr: None if arr is None else arr.ctypes.data_as(C.c_void_p)
        self._check(lib.mrc_encode_stream_pcm16(self._h, n, vp(pl), vp(pr), vp(res_in), vp(res["overall_scale"]),
                                                vp(res.get("ms_switch")), vp(res["scale_factor"]), vp(res["bit_alloc"]),
                                                vp(res["mantissa"]), vp(res["reservoir_out"]), int(chunk_frames)))
        return res

    def encode_stream_pcm16_pac(self, pcm_left, pcm_right=None, reservoir_in=None, use_huffman=True, chunk_frames=0,
                                out=None, bytes_per_chunk=None):
        """mrc_encode_stream_pcm16_pac: int16 PCM stream(s) [(n+1) * L] in host memory -> the `.pac` chunk bytes of the n
        long blocks in host memory (encode kernels + Huffman pricing + bit packing on the device, pipelined).  `out` may hold
        a preallocated (ideally page-locked) uint8 array `bytes`; too small a buffer is retried once at the worst-case size.
        -> dict: bytes (the used prefix), block_offset [n + 1], huff_table / bits_saved [n][channels], reservoir_out [n]."""
        L = self.cfg.n_mdct_lines
        pl = np.ascontiguousarray(pcm_left, dtype=np.int16)
        n = pl.size // L - 1
        if pl.ndim != 1 or pl.size != (n + 1) * L or n < 0:
            raise ValueError("pcm_left must be int16 [(n_frames + 1) * %d]" % L)
        joint = pcm_right is not None
        pr = None
        if joint:
            pr = np.ascontiguousarray(pcm_right, dtype=np.int16)
            if pr.shape != pl.shape:
                raise ValueError("pcm_right must match pcm_left")
        nch = 2 if joint else 1
        bound = int(lib.mrc_pack_bound(C.byref(self.cfg), L, L, 1, 1 if joint else 0))
        buf = None if out is None else out.get("bytes")
        if buf is not None and (buf.dtype != np.uint8 or buf.ndim != 1 or not buf.flags.c_contiguous):
            raise ValueError("out['bytes'] must be a C-contiguous 1-D uint8 array")
        if buf is None:
            buf = np.empty(max(1, n * nch * int(bytes_per_chunk or L) + 4096), np.uint8)
        offs = np.zeros(n + 1, np.int64)
        table, saved = np.zeros((n, nch), np.int32), np.zeros((n, nch), np.int32)
        res_out = np.zeros(n, np.int32)
        res_in = _reservoir(reservoir_in, n)
        total = np.zeros(1, np.int64)
        vp = lambda arr: None if arr is None else arr.ctypes.data_as(C.c_void_p)
        for attempt in (0, 1):
            rc = lib.mrc_encode_stream_pcm16_pac(self._h, n, vp(pl), vp(pr), vp(res_in), 1 if use_huffman else 0, vp(buf),
                                                 buf.size, vp(offs), vp(table), vp(saved), vp(res_out),
                                                 total.ctypes.data_as(_i64p), int(chunk_frames))
            if rc == MRC_ERR_NOMEM and attempt == 0 and buf.size < n * nch * bound:
                buf = np.empty(n * nch * bound, np.uint8)
                continue
            self._check(rc)
            break
        return {"bytes": buf[:int(total[0])], "block_offset": offs, "huff_table": table, "bits_saved": saved,
                "reservoir_out": res_out}

    # ---- chained stream encode (the reference's whole encode loop in one call)
    @staticmethod
    def _chain_schedule(shapes):
        """shapes[s] = [(offset, a, b), ...] (or an int array [n][3]) -> block_start, offset, a, b arrays."""
        if isinstance(shapes, ChainSchedule):
            return shapes.start, shapes.offset, shapes.a, shapes.b
        counts = [len(sh) for sh in shapes]
        start = np.zeros(len(shapes) + 1, np.int64)
        start[1:] = np.cumsum(counts)
        flat = np.concatenate([np.asarray(sh, dtype=np.int64).reshape(-1, 3) for sh in shapes]) if counts and sum(counts) \
            else np.zeros((0, 3), np.int64)
        return (start, np.ascontiguousarray(flat[:, 0]), np.ascontiguousarray(flat[:, 1], dtype=np.int32),
                np.ascontiguousarray(flat[:, 2], dtype=np.int32))

    def encode_chained_pac(self, pcm_left, pcm_right, shapes, use_huffman=True, with_flush=True, num_samples=None,
                           reservoir_in=None, want_trace=False, device=None, stream=None, want_items=False):
        """mrc_encode_chained_stream_pac: stereo streams [nStreams][stride] -- int16 PCM codes or float64 signed fractions,
        each starting with its prior hop -- + the block shapes of every stream -> the `.pac` bytes of every stream (with
        num_samples: complete files, header included), the bit reservoir carried from block to block on the device.
        device = (left_ptr, right_ptr, sample_format, stride, out_ptr, out_cap): everything stays in HBM
        (mrc_dev_encode_chained_pac; `bytes` is then None).
        -> dict: bytes (uint8), stream_offset [nStreams + 1], reservoir_out [nStreams], total, (trace), and with want_items
        item_offset [nItems + 1]: where every block's bytes start (costs a read-back of all chunk positions)."""
        start, off, a, b = self._chain_schedule(shapes)
        n_streams = len(shapes)
        if device is None:
            pl, pr = np.atleast_2d(pcm_left), np.atleast_2d(pcm_right)
            dt = np.int16 if pl.dtype == np.int16 else np.float64
            pl, pr = np.ascontiguousarray(pl, dtype=dt), np.ascontiguousarray(pr, dtype=dt)
            if pl.shape != pr.shape or pl.shape[0] != n_streams:
                raise ValueError("pcm_left / pcm_right must be [nStreams][stride], one row per shape list")
            stride, fmt = pl.shape[1], (1 if dt == np.int16 else 0)
        n_items = len(off) + (2 * n_streams if with_flush else 0)
        ns = None if num_samples is None else np.ascontiguousarray(num_samples, dtype=np.uint32)
        if ns is not None and ns.shape != (n_streams,):
            raise ValueError("num_samples: one value per stream")
        res_in = _reservoir(reservoir_in, n_streams)
        s_off = np.zeros(n_streams + 1, np.int64)
        i_off = np.zeros(n_items + 1, np.int64) if want_items else None
        res_out = np.zeros(n_streams, np.int32)
        trace = np.zeros(n_items, np.int32) if want_trace else None
        total = np.zeros(1, np.int64)
        vp = lambda arr: None if arr is None else arr.ctypes.data_as(C.c_void_p)
        sched = (_p(start, _i64p), _p(off, _i64p), _p(a, _i32p), _p(b, _i32p), _p(res_in, _i32p), 1 if use_huffman else 0,
                 1 if with_flush else 0, vp(ns))
        tail = (_p(s_off, _i64p), _p(i_off, _i64p), _p(res_out, _i32p), _p(trace, _i32p), total.ctypes.data_as(_i64p))
        buf = None
        if device is not None:
            dl, dr, fmt, stride, dout, dcap = device
            self._check(lib.mrc_dev_encode_chained_pac(self._h, n_streams, dl, dr, int(fmt), int(stride), *sched, dout, int(dcap),
                                                       *tail, stream))
        else:
            bound = int(lib.mrc_chain_out_bound(self._h, n_streams, _p(start, _i64p), _p(a, _i32p), _p(b, _i32p),
                                                1 if with_flush else 0, 0 if ns is None else 1))
            if bound < 0:
                raise MrcError("mrc_chain_out_bound failed (%d): block shape out of range" % bound)
            # a buffer for typical content (~3 bits per sample); if the streams pack to more, the call says how much and
            # its bytes -- complete in the handle's device buffer -- are fetched into a buffer of that size (no second encode)
            cap = min(bound, int(off.size) * 1024 + n_streams * 4096 + 4096)
            buf = np.empty(max(cap, 1), np.uint8)
            rc = lib.mrc_encode_chained_stream_pac(self._h, n_streams, vp(pl), vp(pr), fmt, stride, *sched, vp(buf),
                                                   buf.size, *tail)
            if rc == MRC_ERR_NOMEM and 0 < int(total[0]) <= bound:
                buf = np.empty(int(total[0]), np.uint8)
                if lib.mrc_chain_fetch_output(self._h, vp(buf), buf.size, total.ctypes.data_as(_i64p)) != 0:
                    # a call of several slabs keeps only its last slab on the device: encode again into a buffer of the
                    # size the first pass reported
                    self._check(lib.mrc_encode_chained_stream_pac(self._h, n_streams, vp(pl), vp(pr), fmt, stride, *sched,
                                                                  vp(buf), buf.size, *tail))
            else:
                self._check(rc)
            buf = buf[:int(total[0])]
        out = {"bytes": buf, "stream_offset": s_off, "item_offset": i_off, "reservoir_out": res_out, "total": int(total[0])}
        if want_trace:
            out["reservoir_trace"] = trace
        return out

    def chain_ms(self):
        """device time of the last chained encode: phase A + preparation, serial scan, packing, all three (ms)"""
        ms = np.zeros(4, np.float64)
        self._check(lib.mrc_get_chain_ms(self._h, _p(ms, _f64p)))
        return ms

    def pcm_to_float(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.empty(pcm.shape, np.float64)
        self._check(lib.mrc_pcm_to_float(self._h, pcm.size, pcm.ctypes.data_as(C.c_void_p), _p(out, _f64p)))
        return out

    def quantize_uniform(self, x, n_bits):
        x = _f64(np.atleast_1d(x))
        out = np.empty(x.shape, np.int64)
        self._check(lib.mrc_quantize_uniform(self._h, x.size, int(n_bits), _p(x, _f64p), _p(out, _i64p)))
        return out

    def bark(self, f):
        f = _f64(np.atleast_1d(f))
        out = np.empty(f.shape, np.float64)
        self._check(lib.mrc_bark(self._h, f.size, _p(f, _f64p), _p(out, _f64p)))
        return out

    def window(self, blocks, a, b):
        blocks = _f64(blocks)
        out = np.empty_like(blocks)
        self._check(lib.mrc_window(self._h, blocks.shape[0], a, b, _p(blocks, _f64p), _p(out, _f64p)))
        return out

    def mdct(self, blocks, a, b, apply_window=True):
        blocks = _f64(blocks)
        n = blocks.shape[0]
        lines = np.empty((n, (a + b) // 2), np.float64)
        scale = np.empty(n, np.int32)
        self._check(lib.mrc_mdct(self._h, n, a, b, _p(blocks, _f64p), int(bool(apply_window)), _p(lines, _f64p),
                                 _p(scale, _i32p)))
        return lines, scale

    def smr(self, blocks, a, b, scaled_lines=None, overall_scale=None, want_thresh=False):
        blocks = _f64(blocks)
        n = blocks.shape[0]
        nb, half = len(self.bands(a, b)), (a + b) // 2
        smr = np.empty((n, nb), np.float64)
        thr = np.empty((n, half), np.float64) if want_thresh else None
        sl = None if scaled_lines is None else _f64(scaled_lines)
        sc = None if overall_scale is None else _i32(overall_scale)
        self._check(lib.mrc_smr(self._h, n, a, b, _p(blocks, _f64p), _p(sl, _f64p), _p(sc, _i32p), _p(smr, _f64p),
                                _p(thr, _f64p)))
        return (smr, thr) if want_thresh else smr

    def bitalloc(self, budget, max_mant_bits, n_lines, smr, want_smr_after=False):
        """-> (bits [n][nb], bits_left [n]) and, with want_smr_after, the running SMRs the loop leaves behind
        (bitalloc.py:132-151 updates its SMR argument in place)."""
        smr = _f64(np.atleast_2d(smr))
        n, nb = smr.shape
        budget = _f64(np.broadcast_to(np.asarray(budget, dtype=np.float64), (n,)))
        nl = _i32(n_lines)
        bits = np.empty((n, nb), np.int32)
        left = np.empty(n, np.int32)
        if want_smr_after:
            after = smr.copy()
            self._check(lib.mrc_bitalloc_inplace(self._h, n, nb, int(max_mant_bits), _p(nl, _i32p), _p(budget, _f64p),
                                                 _p(after, _f64p), _p(bits, _i32p), _p(left, _i32p)))
            return bits, left, after
        self._check(lib.mrc_bitalloc(self._h, n, nb, int(max_mant_bits), _p(nl, _i32p), _p(budget, _f64p),
                                     _p(smr, _f64p), _p(bits, _i32p), _p(left, _i32p)))
        return bits, left

    def scale_factor(self, v, n_scale_bits, n_mant_bits):
        v = _f64(np.atleast_1d(v))
        mb = _i32(np.broadcast_to(np.asarray(n_mant_bits), v.shape))
        out = np.empty(v.shape, np.int32)
        self._check(lib.mrc_scale_factor(self._h, v.size, int(n_scale_bits), _p(v, _f64p), _p(mb, _i32p),
                                         _p(out, _i32p)))
        return out

    def mantissa(self, x, scale, n_scale_bits, n_mant_bits):
        x = _f64(np.atleast_1d(x))
        sc = _i32(np.broadcast_to(np.asarray(scale), x.shape))
        mb = _i32(np.broadcast_to(np.asarray(n_mant_bits), x.shape))
        out = np.empty(x.shape, np.int32)
        self._check(lib.mrc_mantissa(self._h, x.size, int(n_scale_bits), _p(x, _f64p), _p(sc, _i32p), _p(mb, _i32p),
                                     _p(out, _i32p)))
        return out

    def huffman_gain(self, bit_alloc, mantissa, a, b):
        """bit_alloc [n][nStreams][nBands], mantissa [n][nStreams][N/2] dense -> (table id [n][nStreams], bits_saved)."""
        ba, m = _i32(bit_alloc), _i32(mantissa)
        n, ns = ba.shape[0], ba.shape[1]
        table = np.empty((n, ns), np.int32)
        saved = np.empty((n, ns), np.int32)
        self._check(lib.mrc_huffman_gain(self._h, n, int(a), int(b), ns, _p(ba, _i32p), _p(m, _i32p), _p(table, _i32p),
                                         _p(saved, _i32p)))
        return table, saved

    def dev_huffman_gain(self, a, b, n_frames, n_streams, bit_alloc, mantissa, reservoir_out, huff_table, bits_saved,
                         reservoir_next=None, stream=None):
        self._check(lib.mrc_dev_huffman_gain(self._h, a, b, n_frames, n_streams, bit_alloc, mantissa, reservoir_out,
                                             huff_table, bits_saved, reservoir_next, stream))

    def dev_pack_blocks(self, a, b, n_blocks, n_channels, joint, use_huffman, huff_table_in, overall_scale, ms_switch,
                        scale_factor, bit_alloc, mantissa, mantissa16, out, out_cap, block_offset, huff_table=None,
                        bits_saved=None, want_total=True, stream=None):
        """`.pac` chunks packed on the device (mrc_dev_pack_blocks): every array argument is a DEVICE address.  -> total
        bytes (want_total: waits for the stream), else None."""
        total = np.zeros(1, np.int64)
        self._check(lib.mrc_dev_pack_blocks(self._h, a, b, n_blocks, n_channels, 1 if joint else 0, 1 if use_huffman else 0,
                                            huff_table_in, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa,
                                            1 if mantissa16 else 0, out, out_cap, block_offset, huff_table, bits_saved,
                                            total.ctypes.data_as(_i64p) if want_total else None, stream))
        return int(total[0]) if want_total else None

    # ---- decode side
    def decode(self, a, b, overall_scale, scale_factor, bit_alloc, mantissa, ms_switch=None):
        """codecThem.Decode / JointDecode for n blocks of shape (a,b): scale_factor / bit_alloc [n][nStreams][nBands],
        mantissa [n][nStreams][N/2] dense, overall_scale [n] (mono) or [n][4] with ms_switch [n][nBands] (joint)
        -> windowed blocks [n][nStreams][a+b] (before overlap-and-add)."""
        sf, ba, m, osc = _i32(scale_factor), _i32(bit_alloc), _i32(mantissa), _i32(overall_scale)
        n, ns = sf.shape[0], sf.shape[1]
        nb, half = len(self.bands(a, b)), (a + b) // 2
        if sf.shape != (n, ns, nb) or ba.shape != sf.shape or m.shape != (n, ns, half) or ns not in (1, 2):
            raise ValueError("decode: array shapes do not match the block shape")
        sw = None
        if ns == 2:
            sw = _i32(ms_switch)
            if sw.shape != (n, nb) or osc.shape != (n, 4):
                raise ValueError("decode: joint blocks need overall_scale [n][4] and ms_switch [n][nBands]")
        elif osc.size != n:
            raise ValueError("decode: overall_scale [n] expected")
        out = np.empty((n, ns, a + b), np.float64)
        self._check(lib.mrc_decode(self._h, n, int(a), int(b), ns, _p(osc, _i32p), _p(sw, _i32p), _p(sf, _i32p),
                                   _p(ba, _i32p), _p(m, _i32p), _p(out, _f64p)))
        return out

    def pcm16(self, x):
        x = _f64(x)
        out = np.empty(x.shape, np.int16)
        self._check(lib.mrc_pcm16(self._h, x.size, _p(x, _f64p), out.ctypes.data_as(C.POINTER(C.c_int16))))
        return out

    def dev_decode(self, a, b, n_blocks, n_streams, overall_scale, ms_switch, scale_factor, bit_alloc, mantissa,
                   out_offset, out_left, out_right=None, stream=None):
        self._check(lib.mrc_dev_decode(self._h, a, b, n_blocks, n_streams, overall_scale, ms_switch, scale_factor,
                                       bit_alloc, mantissa, out_offset, out_left, out_right, stream))

    def dev_pcm16(self, n, x, out, stream=None):
        self._check(lib.mrc_dev_pcm16(self._h, n, x, out, stream))

    def transient_peaks(self, streams, sos):
        """streams [nCh][(nHops+1)*hop], float64 signed fractions or the file's int16 PCM codes -> peaks
        [nHops][nCh][hop/nShort + 1] (sub-block peaks, then the hop's peak)."""
        x = np.atleast_2d(streams)
        fmt = 1 if x.dtype == np.int16 else 0
        x = np.ascontiguousarray(x, dtype=np.int16 if fmt else np.float64)
        sos = _f64(sos)
        hop, n_short = self.cfg.n_mdct_lines, self.cfg.n_short
        n_hops = x.shape[1] // hop - 1
        if x.shape[1] != (n_hops + 1) * hop or sos.ndim != 2 or sos.shape[1] != 6:
            raise ValueError("streams must be [nCh][(nHops+1)*hop], sos [nSections][6]")
        out = np.empty((max(n_hops, 0), x.shape[0], hop // n_short + 1), np.float64)
        self._check(lib.mrc_transient_peaks_ex(self._h, n_hops, x.shape[0], sos.shape[0], _p(sos, _f64p),
                                               x.ctypes.data_as(C.c_void_p), fmt, _p(out, _f64p)))
        return out

    def dev_transient_peaks(self, n_hops, n_channels, sos, streams_ptr, sample_format, channel_stride, peaks_ptr, stream=None):
        """mrc_dev_transient_peaks: streams / peaks are device addresses, sos a host array [nSections][6]."""
        sos = _f64(sos)
        self._check(lib.mrc_dev_transient_peaks(self._h, int(n_hops), int(n_channels), sos.shape[0], _p(sos, _f64p), streams_ptr,
                                                int(sample_format), int(channel_stride), peaks_ptr, stream))

    def stereo_masking_factor(self, mid_thresh, side_thresh, z):
        m, s_, zz = _f64(mid_thresh), _f64(side_thresh), _f64(z)
        if not (m.shape == s_.shape == zz.shape):
            raise ValueError("mid, side and z must have the same shape")
        om, os_ = np.empty_like(m), np.empty_like(m)
        self._check(lib.mrc_stereo_masking_factor(self._h, m.size, _p(m, _f64p), _p(s_, _f64p), _p(zz, _f64p),
                                                  _p(om, _f64p), _p(os_, _f64p)))
        return om, os_

    def ms_switch(self, lines_left, lines_right, n_lines):
        L, R = _f64(np.atleast_2d(lines_left)), _f64(np.atleast_2d(lines_right))
        nl = _i32(n_lines)
        if L.shape != R.shape or L.shape[1] != int(nl.sum()):
            raise ValueError("line vectors must be [n][sum(n_lines)]")
        out = np.empty((L.shape[0], len(nl)), np.int32)
        self._check(lib.mrc_ms_switch(self._h, L.shape[0], len(nl), _p(nl, _i32p), _p(L, _f64p), _p(R, _f64p),
                                      _p(out, _i32p)))
        return out

    # ---- device entry points (raw pointers as ints, e.g. torch.Tensor.data_ptr())
    def dev_encode(self, a, b, n_frames, ch_left, ch_right, frame_stride, offsets, reservoir_in, overall_scale,
                   ms_switch, bit_alloc, scale_factor, mantissa, reservoir_out, lines_out=None, stream=None):
        self._check(lib.mrc_dev_encode(self._h, a, b, n_frames, ch_left, ch_right, frame_stride, offsets,
                                       reservoir_in, overall_scale, ms_switch, bit_alloc, scale_factor, mantissa,
                                       reservoir_out, lines_out, stream))

    def dev_encode_ex(self, a, b, n_frames, ch_left, ch_right, sample_format, frame_stride, offsets, reservoir_in,
                      overall_scale, ms_switch, bit_alloc, scale_factor, mantissa, mantissa_format, reservoir_out,
                      lines_out=None, stream=None):
        self._check(lib.mrc_dev_encode_ex(self._h, a, b, n_frames, ch_left, ch_right, int(sample_format), frame_stride,
                                          offsets, reservoir_in, overall_scale, ms_switch, bit_alloc, scale_factor,
                                          mantissa, int(mantissa_format), reservoir_out, lines_out, stream))

    def dev_mdct(self, a, b, n_frames, ch_left, ch_right, frame_stride, offsets, lines, overall_scale, stream=None):
        self._check(lib.mrc_dev_mdct(self._h, a, b, n_frames, ch_left, ch_right, frame_stride, offsets, lines,
                                     overall_scale, stream))

    def dev_smr(self, a, b, n_frames, ch_left, ch_right, frame_stride, offsets, lines, overall_scale, smr,
                thresh=None, stream=None):
        self._check(lib.mrc_dev_smr(self._h, a, b, n_frames, ch_left, ch_right, frame_stride, offsets, lines,
                                    overall_scale, smr, thresh, stream))

    def dev_alloc_quant(self, a, b, n_frames, joint, lines, overall_scale, smr, reservoir_in, ms_switch, bit_alloc,
                        scale_factor, mantissa, reservoir_out, stream=None):
        self._check(lib.mrc_dev_alloc_quant(self._h, a, b, n_frames, int(joint), lines, overall_scale, smr,
                                            reservoir_in, ms_switch, bit_alloc, scale_factor, mantissa,
                                            reservoir_out, stream))

    def set_option(self, option, value):
        self._check(lib.mrc_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int32()
        self._check(lib.mrc_get_option(self._h, int(option), C.byref(v)))
        return v.value

    SENS_NAMES = ("quantiser_edges", "bitalloc_near_ties", "ms_switch_near_threshold", "peak_near_ties",
                  "node_chunks_sent_back", "blocks_examined")

    def sensitivity(self, reset=True):
        """mrc_get_sensitivity: with set_option(5, 1) (MRC_OPT_SENSITIVITY) the encode calls count the integer decisions
        taken within a guard band of floating-point rounding -> dict name -> count since the last reset."""
        v = np.zeros(8, np.int64)
        self._check(lib.mrc_get_sensitivity(self._h, _p(v, _i64p), 1 if reset else 0))
        return {n: int(v[i]) for i, n in enumerate(self.SENS_NAMES)}

    def set_timing(self, on):
        self._check(lib.mrc_set_timing(self._h, int(bool(on))))

    def stage_ms(self):
        ms = np.zeros(3, np.float64)
        self._check(lib.mrc_get_stage_ms(self._h, _p(ms, _f64p)))
        return ms

    def kernel_ms(self):
        """device time of the last timed encode per kernel: MDCT, smr, band_stats (joint), bitalloc, quantize"""
        ms = np.zeros(5, np.float64)
        self._check(lib.mrc_get_kernel_ms(self._h, _p(ms, _f64p)))
        return ms


MRC_SAMPLES_F64, MRC_SAMPLES_PCM16 = 0, 1
MRC_MANTISSA_I32, MRC_MANTISSA_I16 = 0, 1


class PinnedArray:
    """A NumPy array over page-locked host memory (mrc_host_alloc): what makes the copies of
    Handle.encode_stream_pcm16 asynchronous.  Keep the object alive as long as `.array` is used."""

    def __init__(self, shape, dtype):
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        self._p = C.c_void_p()
        rc = lib.mrc_host_alloc(C.byref(self._p), max(n, 1))
        if rc != 0:
            raise MrcError("mrc_host_alloc(%d bytes) failed (%d)" % (n, rc))
        buf = (C.c_char * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.array = None
            lib.mrc_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

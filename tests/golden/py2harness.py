"""
Run the reference's OWN function bodies under CPython 3 with their Python-2 / NumPy<1.12 meaning.

Used only by tests/golden/make_golden_ref.py, in the build container (the reference never travels); the
outputs are committed as data fixtures (tests/golden/ref_*.npz).  Nothing of the reference's text is stored.

Why a harness: the reference is Python 2.  Under Python 3.10 / NumPy 2.2
  * mdct.py does not parse -- but only its `if __name__ == "__main__":` self-test (mdct.py:127-213, py2
    `print` statements) is at fault; the function definitions above it (mdct.py:1-126) are valid Python 3;
  * psychoac.py, window.py, codecThem.py parse, but rely on Python-2 integer `/` (psychoac.py:160,165,
    codecThem.py:470, mdct.py:70,75), on float-valued sizes / slice bounds (window.py:60,79,90,
    codecThem.py:288,317,336-340) that NumPy < 1.12 truncated to int, and on `dict.has_key`
    (codecThem.py:169,194).
The harness loads each module from its source text, applies ONE uniform AST pass that restores those
meanings, and executes the result.  The pass rewrites no logic:
  * `a / b`, `a /= b`      -> classic division: floor division when both operands are integers
                              (Python ints, NumPy integer scalars / arrays), true division otherwise;
  * `x[i]`, `x[i:j:k]`     -> float index / slice bounds with an integral value become ints (asserted integral);
  * `range`, `np.zeros`, `np.ones`, `np.empty`, `np.linspace(num=)` accept such floats too (module globals);
  * `np.right_shift` / `np.left_shift(array, numpy_integer_scalar)` take the scalar by value, as NumPy < 2
    did (quantize.py:316,344);
  * mdct.py is cut at its `__main__` self-test line; no other text is dropped.
What this cannot reproduce is the last-ulp behaviour of the NumPy build the authors used (its FFT was
FFTPACK, ours is pocketfft; libm differs): fixtures are compared bit-exactly for integers and with a stated
float tolerance otherwise.
"""
import ast
import builtins
import os
import sys
import types

import numpy as np

_INT_TYPES = (int, np.integer)


def _is_int(v):
    if isinstance(v, bool):
        return True
    if isinstance(v, _INT_TYPES):
        return True
    if isinstance(v, np.ndarray) and v.dtype.kind in "iub":
        return True
    return False


def py2div(a, b):
    """Python 2 classic division."""
    if _is_int(a) and _is_int(b):
        return a // b
    return a / b


def py2idiv(a, b):
    """`a /= b` with Python 2 meaning: in place for ndarrays (as NumPy does), rebinding otherwise."""
    if isinstance(a, np.ndarray):
        if _is_int(a) and _is_int(b):
            a //= b
        else:
            a /= b
        return a
    return py2div(a, b)


def py2idx(v):
    """An index / slice bound / size as NumPy < 1.12 took it: floats are truncated (here: must be integral)."""
    if isinstance(v, tuple):
        return tuple(py2idx(e) for e in v)
    if isinstance(v, (float, np.floating)):
        assert float(v) == int(v), "non-integral float used as an index/size: %r" % (v,)
        return int(v)
    return v


def py2range(*args):
    return builtins.range(*[py2idx(a) for a in args])


class _Np:
    """numpy, with the size arguments of the allocation helpers taken as NumPy < 1.12 did."""

    float, int, bool = float, int, bool      # aliases NumPy < 1.24 had (pacfileThem.py:981 uses np.float)

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def zeros(shape, *a, **k):
        return np.zeros(py2idx(shape), *a, **k)

    @staticmethod
    def ones(shape, *a, **k):
        return np.ones(py2idx(shape), *a, **k)

    @staticmethod
    def empty(shape, *a, **k):
        return np.empty(py2idx(shape), *a, **k)

    @staticmethod
    def linspace(start, stop, num=50, *a, **k):
        return np.linspace(start, stop, py2idx(num), *a, **k)

    @staticmethod
    def fromstring(data, dtype=float, *a, **k):
        # bitpack.py:34 (decoder side): byte string -> uint8 array.  Its elements then meet Python ints in
        # `mask & data[i]`, `x << n`; NumPy < 2 promoted uint8-scalar (op) int to int64, NumPy 2 keeps uint8 and the
        # shifts wrap.  The array is handed out as int64 so that the arithmetic has its old range.
        arr = np.frombuffer(_b(data), dtype=dtype)
        return arr.astype(np.int64) if arr.dtype == np.uint8 else arr.copy()

    @staticmethod
    def right_shift(a, b, *r, **k):
        # NumPy < 2 cast SCALAR operands by value: uint64-array >> np.int32(3) was legal (quantize.py:316 gets
        # its shift count from an int32 array element); NumPy 2 (NEP 50) refuses the uint64/int32 pair.
        if isinstance(b, np.integer):
            b = int(b)
        return np.right_shift(a, b, *r, **k)

    @staticmethod
    def left_shift(a, b, *r, **k):        # same, quantize.py:344
        if isinstance(b, np.integer):
            b = int(b)
        return np.left_shift(a, b, *r, **k)


class Py2Dict(dict):
    """dict with Python 2's has_key (codecThem.py:169,194) -- used for the Huffman tables WE write out."""

    def has_key(self, k):
        return k in self


def _name(n):
    return ast.Name(id=n, ctx=ast.Load())


class _Pass(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=_name("__py2div__"), args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            load = ast.parse(ast.unparse(node.target), mode="eval").body
            new = ast.Assign(targets=[node.target],
                             value=ast.Call(func=_name("__py2idiv__"), args=[load, node.value], keywords=[]))
            return ast.copy_location(new, node)
        return node

    def visit_Subscript(self, node):
        self.generic_visit(node)

        def wrap(e):
            if e is None:
                return None
            return ast.Call(func=_name("__py2idx__"), args=[e], keywords=[])

        s = node.slice
        if isinstance(s, ast.Slice):
            node.slice = ast.Slice(lower=wrap(s.lower), upper=wrap(s.upper), step=wrap(s.step))
        elif isinstance(s, ast.Tuple):
            node.slice = ast.Tuple(elts=[ast.Slice(lower=wrap(e.lower), upper=wrap(e.upper), step=wrap(e.step))
                                         if isinstance(e, ast.Slice) else wrap(e) for e in s.elts], ctx=ast.Load())
        else:
            node.slice = wrap(s)
        return node


def _load(name, ref_dir, cut_main=False):
    path = os.path.join(ref_dir, name + ".py")
    with open(path, encoding="utf-8-sig") as f:
        text = f.read()
    if cut_main:
        lines = text.splitlines(keepends=True)
        for i, ln in enumerate(lines):
            if ln.startswith("if __name__"):
                lines = lines[:i]
                break
        text = "".join(lines)
    tree = _Pass().visit(ast.parse(text, filename=path))
    ast.fix_missing_locations(tree)
    mod = types.ModuleType(name)
    mod.__file__ = path
    mod.__dict__.update(__py2div__=py2div, __py2idiv__=py2idiv, __py2idx__=py2idx, range=py2range, xrange=py2range)
    sys.modules[name] = mod
    exec(compile(tree, path, "exec"), mod.__dict__)
    if "np" in mod.__dict__:
        mod.np = _Np()
    mod.range = py2range          # `from x import *` may have rebound it
    return mod


ORDER = ("window", "mdct", "quantize", "bitalloc", "ms_stereo", "psychoac", "codecThem")


def load_reference(ref_dir="/root/reference"):
    """-> dict name -> module, the reference's hot-path modules executed with their Python 2 meaning."""
    sys.dont_write_bytecode = True
    mods = {}
    for name in ORDER:
        mods[name] = _load(name, ref_dir, cut_main=(name == "mdct"))
    return mods


# ---------------------------------------------------------------------------------------------- file layer
# pacfileThem.py, bitpack.py, huffman.py additionally contain Python 2 `print` STATEMENTS inside functions; they are
# rewritten to print() calls, in memory, by lib2to3's print fixer (no other fixer runs).  Python 2 `str` is a byte
# string: the file objects the reference opens through audiofile.py and the struct helpers it calls are given that
# meaning here (bytes <-> str through latin-1, a bijection), so `tag == "RIFF"`, `"\0" * n` and
# `fp.write(self.tag)` behave as they did.
class _ByteStrFile:
    """open(name, 'rb'/'wb') whose read() returns / write() accepts Python-2 style byte strings."""

    def __init__(self, name, mode="r"):
        self._f = builtins.open(name, mode if "b" in mode else mode + "b")
        self.mode = mode
        self.name = name

    def read(self, n=-1):
        return self._f.read(n).decode("latin-1")

    def write(self, s):
        if isinstance(s, str):
            s = s.encode("latin-1")
        return self._f.write(s)

    def seek(self, *a):
        return self._f.seek(*a)

    def tell(self):
        return self._f.tell()

    def close(self):
        return self._f.close()


def _b(v):
    return v.encode("latin-1") if isinstance(v, str) else v


def _struct_helpers():
    import struct

    def pack(fmt, *args):
        return struct.pack(fmt, *[_b(a) for a in args]).decode("latin-1")

    def unpack(fmt, s):
        return tuple(v.decode("latin-1") if isinstance(v, bytes) else v for v in struct.unpack(fmt, _b(s)))

    return dict(pack=pack, unpack=unpack, calcsize=struct.calcsize)


def _print_fixed(text, name):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
        return str(tool.refactor_string(text if text.endswith("\n") else text + "\n", name))


FILE_ORDER = ("audiofile", "bitpack", "huffman", "pcmfile")


def load_file_layer(ref_dir="/root/reference"):
    """-> dict of the file-layer modules (audiofile, bitpack, huffman, pcmfile) on top of load_reference()."""
    mods = load_reference(ref_dir)
    import queue
    sys.modules.setdefault("Queue", queue)          # Python 2 name of the module (huffman.py:3)
    for name in FILE_ORDER:
        path = os.path.join(ref_dir, name + ".py")
        with builtins.open(path, encoding="utf-8-sig") as f:
            text = _print_fixed(f.read(), name)
        tree = _Pass().visit(ast.parse(text, filename=path))
        ast.fix_missing_locations(tree)
        mod = types.ModuleType(name)
        mod.__file__ = path
        mod.__dict__.update(__py2div__=py2div, __py2idiv__=py2idiv, __py2idx__=py2idx, range=py2range, xrange=py2range)
        sys.modules[name] = mod
        exec(compile(tree, path, "exec"), mod.__dict__)
        _post(mod)
        mods[name] = mod
    return mods


def _post(mod):
    if "np" in mod.__dict__:
        mod.np = _Np()
    mod.range = py2range
    for k, v in _struct_helpers().items():
        if k in mod.__dict__:
            mod.__dict__[k] = v
    if mod.__name__ == "audiofile":
        mod.open = _ByteStrFile


def _open_binary(name, mode="r", *a, **k):
    """Python 2 on Linux: text mode == binary mode (pacfileThem.py:247 opens a pickle with mode 'r')."""
    return builtins.open(name, mode if "b" in mode else mode + "b", *a, **k)


def write_huffman_files(directory, tables, order):
    """Write the three files per table the reference's encoder and decoder look for under ./training_data
    (`*table.pkl` codecThem.py:137-138; `*tree.pkl`, `*_table.revpkl` pacfileThem.py:170-171,246), from code
    tables given as data: tables[name] = ({value: (code string, length)}, escape value).  The reference numbers
    tables by os.walk/glob order, which is filesystem dependent and differs between the three patterns
    (SURVEY F9): empty files are created first, the order each pattern is FOUND in is observed, and table
    order[i] is written into the i-th file found, so that every index means the same table.
    Trees are built from the reference's own HuffmanNode class (huffman.py:5-9): children are (child, weight)
    tuples, '0' = left, '1' = right, exactly what its createTree (huffman.py:99-110) nests."""
    import pickle
    from glob import glob
    hm = sys.modules["huffman"]
    for i in range(len(order)):
        d = os.path.join(directory, "t%d" % i)
        os.makedirs(d)
        for suffix in ("_table.pkl", "_tree.pkl", "_table.revpkl"):
            builtins.open(os.path.join(d, "t%d%s" % (i, suffix)), "wb").close()

    def found(pattern):
        out = [y for x in os.walk(directory) for y in glob(os.path.join(x[0], pattern))]
        assert len(out) == len(order), (pattern, out)
        return out

    def tree_of(table):
        def build(prefix):
            kids = []
            for bit in "01":
                code = prefix + bit
                leaf = [v for v, (c, _n) in table.items() if c == code]
                if leaf:
                    kids.append((leaf[0], 1))
                elif any(c.startswith(code) for (c, _n) in table.values()):
                    kids.append((build(code), 1))
                else:
                    kids.append(None)
            return hm.HuffmanNode(kids[0], kids[1])
        return (build(""), 1)

    for path, name in zip(found("*table.pkl"), order):
        table, escape = tables[name]
        with builtins.open(path, "wb") as f:
            pickle.dump((Py2Dict((int(v), (str(c), int(n))) for v, (c, n) in table.items()), int(escape)), f, protocol=2)
    for path, name in zip(found("*tree.pkl"), order):
        table, escape = tables[name]
        with builtins.open(path, "wb") as f:
            pickle.dump((tree_of(table), int(escape)), f, protocol=2)
    for path, name in zip(found("*_table.revpkl"), order):
        table, escape = tables[name]
        with builtins.open(path, "wb") as f:
            pickle.dump((Py2Dict((str(c), int(v)) for v, (c, n) in table.items()), str(table[escape][0])), f, protocol=2)


def run_pacfile_main(wav_path, ref_dir="/root/reference"):
    """Execute pacfileThem.py AS A SCRIPT (its `if __name__ == "__main__":` driver, pacfileThem.py:1064-1231) on
    wav_path, in the current directory (./training_data is looked up relative to it).  Returns the exception the
    script ended with, or None."""
    load_file_layer(ref_dir)
    path = os.path.join(ref_dir, "pacfileThem.py")
    with builtins.open(path, encoding="utf-8-sig") as f:
        text = _print_fixed(f.read(), "pacfileThem")
    tree = _Pass().visit(ast.parse(text, filename=path))
    ast.fix_missing_locations(tree)
    g = dict(__name__="__main__", __file__=path, __py2div__=py2div, __py2idiv__=py2idiv, __py2idx__=py2idx,
             range=py2range, xrange=py2range)
    code = compile(tree, path, "exec")
    argv = sys.argv
    sys.argv = [path, wav_path]

    class _Globals(dict):
        """module namespace that re-applies the helpers after the script's own imports rebind them"""
        def __setitem__(self, k, v):
            if k == "np":
                v = _Np()
            elif k == "range":
                v = py2range
            elif k in ("pack", "unpack", "calcsize"):
                v = _struct_helpers()[k]
            elif k == "open":
                v = _open_binary         # `from audiofile import *` must not leak the byte-string file class
            dict.__setitem__(self, k, v)

    ns = _Globals(g)
    err = None
    try:
        exec(code, ns)
    except BaseException as e:          # the decode half needs files this harness does not provide
        err = e
    finally:
        sys.argv = argv
    return err

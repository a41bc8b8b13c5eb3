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

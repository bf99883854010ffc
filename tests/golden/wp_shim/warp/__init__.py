"""A numpy interpreter of the subset of NVIDIA Warp 0.10.1 that PhysGaussian's MPM solver uses -- TEST INFRASTRUCTURE ONLY.

Purpose: pin the MPM oracle to the reference's own code.  Warp is not installed in the build container (and has no ROCm
back end), but the reference's kernels are plain Python source (third_party/PhysGaussian/mpm_solver_warp/{mpm_utils,
warp_utils,mpm_solver_warp}.py).  With this package first on sys.path, `import mpm_solver_warp` imports the REFERENCE'S
FILES UNMODIFIED and `MPM_Simulator_WARP` runs as written: `wp.launch` loops over the launch grid and calls the kernel's
Python body once per thread.  tests/golden/make_mpm_ref_golden.py drives it and commits the fixture.

What this file has to get right is Warp's *typing*, because the kernels are written against it:

  * a float literal inside a @wp.kernel / @wp.func is a float32 constant.  The decorators therefore re-compile the
    function from its source (as Warp's own code generator does) with every float literal passed through float32;
  * `float` members of a @wp.struct, Python floats given to wp.launch and vec3/mat33 built in host scope are float32;
  * built-ins called from host scope (wp.sin / wp.sqrt in mpm_solver_warp.py:91-92, :392-393, :760, :1092) go through
    Warp's C library with float32 arguments and results;
  * `wp.mat33(v0, v1, v2)` with three vectors takes them as COLUMNS (w[axis, offset] in mpm_utils.py:350-358 only sums
    to one under that reading);  `mat * mat` is the matrix product, `mat * vec` the matrix-vector product;
  * `wp.int(x)` truncates toward zero;  unset struct members are zero (model.hardening, model.xi ...);
  * `wp.svd3(A, U, s, V)` fills U, s, V in place.  It is the ONE built-in that stays a stand-in (its source is Warp's
    native svd.h, which is not in the reference tree): LAPACK's SVD canonicalised to the convention of Warp's
    implementation -- U, V proper rotations, s0 >= s1 >= |s2|, the sign of det A carried by s2.  WP_SHIM_SVD=lapack
    keeps LAPACK's own convention (s >= 0, U or V improper when det A < 0) so the fixture can show what depends on it.

Arithmetic precision: WP_SHIM_PRECISION=f64 (default) keeps the float32 *problem data* above but evaluates every
kernel expression in double -- the float64 oracle build (oracle/mpm_oracle.c, -DREAL=double) follows the same rule, so the
two can be compared to ~1e-12.  WP_SHIM_PRECISION=f32 evaluates in numpy float32 (what the Warp kernels do up to the
last-bit behaviour of CUDA's logf/expf and fused multiply-adds).
"""
import ast
import builtins as _b
import inspect
import itertools
import numbers
import os
import textwrap

import numpy as np

from . import types  # noqa: F401  (warp.types.array / warp.types.float32 are used by warp_utils.py:246-330)

PRECISION = os.environ.get("WP_SHIM_PRECISION", "f64")
DT = np.float64 if PRECISION == "f64" else np.float32
SVD_CONVENTION = os.environ.get("WP_SHIM_SVD", "warp")

_tid = None          # current thread index while a kernel body runs; None in host scope


def set_precision(name):
    """Switch the arithmetic type ("f64" / "f32") for solvers created from now on."""
    global PRECISION, DT
    PRECISION, DT = name, {"f64": np.float64, "f32": np.float32}[name]


def set_svd_convention(name):
    global SVD_CONVENTION
    assert name in ("warp", "lapack")
    SVD_CONVENTION = name


def _in_kernel():
    return _tid is not None


def _f32(x):
    """A host float as the float32 a kernel / struct / vector receives, kept in the arithmetic type."""
    return DT(np.float32(x))


def _scalar(x):
    return isinstance(x, (numbers.Number, np.generic))


def _store(a):
    """Vector / matrix storage rule: float32 in host scope (Warp's vec3/mat33 ARE float32), the arithmetic type inside
    a kernel (where the float64 mode keeps double precision on purpose)."""
    a = np.asarray(a)
    if _in_kernel():
        return a.astype(DT, copy=True)
    return a.astype(np.float32).astype(DT)


# ----------------------------------------------------------------------------- vectors and matrices
class _Vec:
    N = 3
    __slots__ = ("a",)
    __array_ufunc__ = None      # numpy scalars defer to __rmul__ instead of iterating the vector

    def __init__(self, *args):
        n = self.N
        if len(args) == 0:
            a = np.zeros(n)
        elif len(args) == 1 and _scalar(args[0]):
            a = np.full(n, args[0], dtype=np.float64 if not isinstance(args[0], np.floating) else None)
        elif len(args) == 1:
            src = args[0].a if isinstance(args[0], _Vec) else args[0]
            a = np.array([_b.float(v) for v in src]) if not isinstance(src, np.ndarray) else src
            assert a.shape == (n,), a.shape
        else:
            assert len(args) == n, args
            a = np.array(args)
        self.a = _store(a)

    @classmethod
    def _wrap(cls, a):
        v = cls.__new__(cls)
        v.a = _store(a)
        return v

    def __getitem__(self, i):
        return self.a[i]

    def __setitem__(self, i, val):
        self.a[i] = val

    def __len__(self):
        return self.N

    def __iter__(self):
        return iter(self.a)

    def __add__(self, o):
        assert isinstance(o, _Vec)
        return self._wrap(self.a + o.a)

    def __sub__(self, o):
        assert isinstance(o, _Vec)
        return self._wrap(self.a - o.a)

    def __neg__(self):
        return self._wrap(-self.a)

    def __mul__(self, s):
        assert _scalar(s), "vec * vec is not defined in Warp (cw_mul)"
        return self._wrap(self.a * s)

    __rmul__ = __mul__

    def __truediv__(self, s):
        assert _scalar(s)
        return self._wrap(self.a / s)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(_b.float(x)) for x in self.a)})"


class vec2(_Vec):
    N = 2
    __slots__ = ()


class vec3(_Vec):
    N = 3
    __slots__ = ()


class quat(_Vec):
    N = 4
    __slots__ = ()


class mat33:
    __slots__ = ("a",)
    __array_ufunc__ = None

    def __init__(self, *args):
        if len(args) == 0:
            a = np.zeros((3, 3))
        elif len(args) == 1 and _scalar(args[0]):
            a = np.full((3, 3), _b.float(args[0]))        # wp.mat33(0.0): every element (mpm_utils.py:316, 535, 562)
        elif len(args) == 3 and all(isinstance(v, _Vec) for v in args):
            a = np.stack([v.a for v in args], axis=1)  # three vectors = three COLUMNS
        elif len(args) == 9:
            a = np.array(args).reshape(3, 3)           # nine scalars, row-major
        elif len(args) == 1 and isinstance(args[0], mat33):
            a = args[0].a
        else:
            raise TypeError(f"mat33{args}")
        self.a = _store(a)

    @classmethod
    def _wrap(cls, a):
        m = cls.__new__(cls)
        m.a = _store(a)
        return m

    def __getitem__(self, ij):
        return self.a[ij]

    def __setitem__(self, ij, val):
        self.a[ij] = val

    def __add__(self, o):
        assert isinstance(o, mat33)
        return self._wrap(self.a + o.a)

    def __sub__(self, o):
        assert isinstance(o, mat33)
        return self._wrap(self.a - o.a)

    def __neg__(self):
        return self._wrap(-self.a)

    def __mul__(self, o):
        if isinstance(o, mat33):
            return self._wrap(_matmul(self.a, o.a))
        if isinstance(o, vec3):
            return vec3._wrap(_matvec(self.a, o.a))
        assert _scalar(o)
        return self._wrap(self.a * o)

    def __rmul__(self, s):
        assert _scalar(s)
        return self._wrap(self.a * s)

    def __truediv__(self, s):
        assert _scalar(s)
        return self._wrap(self.a / s)

    def __repr__(self):
        return f"mat33({self.a.tolist()})"


def _matmul(A, B):
    """3x3 product with the summation order of the obvious triple loop (k = 0, 1, 2), no BLAS, no FMA."""
    out = np.empty((3, 3), dtype=np.result_type(A, B))
    for i in range(3):
        for j in range(3):
            out[i, j] = A[i, 0] * B[0, j] + A[i, 1] * B[1, j] + A[i, 2] * B[2, j]
    return out


def _matvec(A, v):
    out = np.empty(3, dtype=np.result_type(A, v))
    for i in range(3):
        out[i] = A[i, 0] * v[0] + A[i, 1] * v[1] + A[i, 2] * v[2]
    return out


# ----------------------------------------------------------------------------- scalar built-ins
def _num(x):
    """Scalar argument of a built-in: the arithmetic type inside a kernel, float32 in host scope."""
    return DT(x) if _in_kernel() else np.float32(x)


def _ret(x):
    return x if _in_kernel() else _b.float(x)


def _unary(fn):
    def call(x):
        return _ret(fn(_num(x)))
    return call


log, exp, sqrt, sin, cos, acos = (_unary(f) for f in (np.log, np.exp, np.sqrt, np.sin, np.cos, np.arccos))


def pow(x, y):  # noqa: A001
    return _ret(np.power(_num(x), _num(y)))


def abs(x):  # noqa: A001
    return -x if x < 0 else x


def max(a, b):  # noqa: A001
    return a if a > b else b


def min(a, b):  # noqa: A001
    return a if a < b else b


def int(x):  # noqa: A001
    return _b.int(x)   # C cast: toward zero


def float(x):  # noqa: A001
    return DT(x)


float32 = np.float32


# ----------------------------------------------------------------------------- vector / matrix built-ins
def dot(a, b):
    return a.a[0] * b.a[0] + a.a[1] * b.a[1] + a.a[2] * b.a[2]


def length(v):
    d = dot(v, v)
    return np.sqrt(d) if _in_kernel() else _b.float(np.sqrt(np.float32(d)))


def normalize(v):
    return v / length(v)


def cross(a, b):
    a, b = a.a, b.a
    return vec3._wrap(np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]))


def cw_mul(a, b):
    return type(a)._wrap(a.a * b.a)


def add(a, b):
    return a + b


def sub(a, b):
    return a - b


def transpose(m):
    return mat33._wrap(m.a.T)


def determinant(m):
    a = m.a
    return (a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
            + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))


def outer(a, b):
    return mat33._wrap(np.multiply.outer(a.a, b.a))


def diag(v):
    return mat33._wrap(np.diag(v.a))


def svd3(A, U, sigma, V):
    """Stand-in for Warp's native 3x3 SVD (see the module docstring); fills U, sigma, V in place."""
    u, s, vt = np.linalg.svd(A.a.astype(np.float64))
    v = vt.T
    if SVD_CONVENTION == "warp":
        if np.linalg.det(u) < 0:
            u[:, 2] = -u[:, 2]; s[2] = -s[2]
        if np.linalg.det(v) < 0:
            v[:, 2] = -v[:, 2]; s[2] = -s[2]
    U.a[...] = u.astype(DT)
    sigma.a[...] = s.astype(DT)
    V.a[...] = v.astype(DT)


# ----------------------------------------------------------------------------- arrays
_ELEM = {vec3: (3,), mat33: (3, 3), vec2: (2,), quat: (4,)}
_py_int = type(1)
_py_float = type(1.0)


class array:
    """wp.array: `dtype` in {float, int, vec3, mat33}; indexing returns VALUES (a read of a vec3/mat33 element is a copy,
    as in a kernel)."""

    def __init__(self, data=None, dtype=None, **_):
        self.dtype = dtype
        self.data = data
        if data is not None:
            self.shape = data.shape[:data.ndim - len(_ELEM.get(dtype, ()))]

    def __getitem__(self, idx):
        val = self.data[idx]
        if self.dtype is vec3:
            return vec3._wrap(val)
        if self.dtype is mat33:
            return mat33._wrap(val)
        return val

    def __setitem__(self, idx, val):
        if isinstance(val, (_Vec, mat33)):
            val = val.a
        self.data[idx] = val

    def numpy(self):
        return self.data.copy()

    def __len__(self):
        return self.shape[0]


def _np_dtype(dtype):
    if dtype in (_py_int, np.int32):
        return np.int32
    return DT


def zeros(shape, dtype=_py_float, device=None, **_):
    shape = (shape,) if isinstance(shape, numbers.Integral) else tuple(shape)
    return array(np.zeros(shape + _ELEM.get(dtype, ()), dtype=_np_dtype(dtype)), dtype)


empty = zeros


def from_numpy(arr, dtype=_py_float, device=None, **_):
    arr = np.asarray(arr)
    nd = _np_dtype(dtype)
    data = arr.astype(np.float32).astype(nd) if nd is not np.int32 else arr.astype(np.int32)   # Warp arrays are float32
    return array(np.ascontiguousarray(data), dtype)


def from_torch(t, dtype=None):
    a = t.detach().cpu().numpy()
    if dtype is None:
        dtype = _py_int if a.dtype.kind in "iu" else (vec3 if a.ndim == 2 and a.shape[1] == 3 else _py_float)
    return from_numpy(a, dtype)


def to_torch(arr):
    import torch
    return torch.from_numpy(arr.data)      # aliases the array, as wp.to_torch does


def atomic_add(arr, *args):
    *idx, val = args
    for i, n in zip(idx, arr.shape):
        assert 0 <= i < n, ("atomic_add outside the grid", idx, arr.shape)
    if isinstance(val, _Vec):
        val = val.a
    arr.data[tuple(idx)] += val


# ----------------------------------------------------------------------------- structs
def struct(cls):
    ann = dict(cls.__annotations__)

    def __init__(self):
        for name, t in ann.items():
            if t is _py_float:
                object.__setattr__(self, name, DT(0.0))
            elif t is _py_int:
                object.__setattr__(self, name, 0)
            elif t in (vec3, vec2):
                object.__setattr__(self, name, t())
            else:
                object.__setattr__(self, name, None)

    def __setattr__(self, name, val):
        t = ann.get(name)
        if t is _py_float:
            val = _f32(val)
        elif t is _py_int:
            val = _py_int(val)
        elif t in (vec3, vec2) and not isinstance(val, t):
            val = t(*val)
        object.__setattr__(self, name, val)

    return type(cls.__name__, (), {"__init__": __init__, "__setattr__": __setattr__, "_fields": ann})


# ----------------------------------------------------------------------------- kernels
class _Float32Literals(ast.NodeTransformer):
    def visit_Constant(self, node):
        if isinstance(node.value, _py_float):
            return ast.copy_location(ast.Call(func=ast.Name(id="__wp_lit__", ctx=ast.Load()), args=[node], keywords=[]), node)
        return node


def _lit(x):
    return DT(np.float32(x))


def _recompile(fn):
    """Re-compile a kernel / device function from its source with float32 literals (what Warp's code generator does)."""
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    fdef = tree.body[0]
    assert isinstance(fdef, ast.FunctionDef) and fdef.name == fn.__name__
    fdef.decorator_list = []
    for a in fdef.args.args:
        a.annotation = None
    fdef.returns = None
    tree = ast.fix_missing_locations(_Float32Literals().visit(tree))
    ast.increment_lineno(tree, fn.__code__.co_firstlineno - 1)      # tracebacks point at the reference's own lines
    code = compile(tree, inspect.getsourcefile(fn), "exec")
    if fn.__closure__:       # kernels defined inside a method capture host values (translation_scale, mpm_solver_warp.py:1165)
        glb = dict(fn.__globals__)
        for name, cell in zip(fn.__code__.co_freevars, fn.__closure__):
            val = cell.cell_contents
            glb[name] = _f32(val) if isinstance(val, _py_float) else val
    else:
        glb = fn.__globals__  # live: device functions defined later in the module resolve at call time
    glb["__wp_lit__"] = _lit
    scratch = {}
    exec(code, glb, scratch)
    return scratch[fn.__name__]


def func(fn):
    return _recompile(fn)


class Kernel:
    def __init__(self, fn):
        self.name = fn.__name__
        self.body = _recompile(fn)


def kernel(fn):
    return Kernel(fn)


def tid():
    return _tid


LAUNCH_LOG = []      # (kernel name, dim) of every launch, so a driver can check the order of p2g2p


def launch(kernel, dim, inputs=(), device=None, **_):   # noqa: A002
    global _tid
    assert isinstance(kernel, Kernel)
    LAUNCH_LOG.append((kernel.name, dim))
    args = [_f32(a) if isinstance(a, (_py_float, np.floating)) else a for a in inputs]
    try:
        if isinstance(dim, numbers.Integral):
            for p in range(dim):
                _tid = p
                kernel.body(*args)
        else:
            for t in itertools.product(*(range(d) for d in dim)):
                _tid = t
                kernel.body(*args)
    finally:
        _tid = None


class ScopedTimer:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def synchronize():
    pass


def init():
    pass

"""A NumPy interpreter of the Taichi subset that third_party/PhysGaussian/particle_filling/filling.py uses -- TEST
INFRASTRUCTURE ONLY (tests/golden/make_filling_ref_golden.py and the live re-computation test import it; pixie_amd/ never
does).  Taichi is not installed in this image; its kernels are Python source, so the reference's own filling.py can be
imported UNMODIFIED on top of this module and executed statement by statement.

What is emulated, and how (Taichi 1.x semantics, `ti.init(arch=ti.cuda)` defaults: default_fp = f32, default_ip = i32):
  * @ti.kernel / @ti.func: re-compiled from their source with every float literal turned into the working precision
    (float32 by default; `set_precision("f64")` runs the same source in float64 -- the fixture's reference values -- so the
    float32 run measures the rounding of the reference's own arithmetic).  Kernel arguments annotated `float` / `int` are
    cast at the call (to float32 in both modes: the problem data are the reference's, only the arithmetic is widened),
    `ti.template()` arguments (fields) are passed through.
  * the outermost `for` of a kernel is a parallel loop in Taichi; here it runs serially in index order, which is one of the
    schedules Taichi may take.  The kernels of filling.py only ever write the cell / particle they iterate over (plus
    `ti.atomic_add`), so every schedule gives the same grids; the ORDER in which new particles land in the output buffer
    and their `ti.random()` offsets are schedule- and generator-dependent in the reference itself and are not pinned.
  * integers are Python ints, floats are NumPy scalars of the working precision: int (op) float gives the float type, as
    Taichi's i32 (op) f32 -> f32.  `ti.floor(x, dtype=int)` / `ti.ceil(x, dtype=int)` return ints; `range(a, b)` casts its
    bounds to int (Taichi casts range bounds to i32 -- filling.py:68-71 relies on it: `r` is a float variable there).
  * ti.Vector / ti.Matrix: small dense values with value semantics (`index += dir` rebinds, fields hand out copies);
    `a.dot(b)`, `A @ v`, `A @ B`, `.transpose()`, `.norm()` accumulate left to right in the working precision.
  * ti.atomic_add(x, v) returns the old value; its first argument is an lvalue, so the call is rewritten on the AST
    (subscript target -> field method; local name -> rebinding expression).
  * ti.field / ti.Vector.field: NumPy arrays with from_torch / to_torch / from_numpy / to_numpy; struct-for
    (`for i, j, k in field`) iterates all indices in C order; a ti.Vector of ints indexes a field.
  * ti.sym_eig is the ONE stand-in (like wp.svd3 in tests/golden/wp_shim): LAPACK's symmetric eigensolver in the working
    precision.  filling.py:51-59 only uses it to form Q diag(1/max(sig, 1e-8)) Q^T and max(sig), which do not depend on the
    order or the signs of the eigenvectors.
  * ti.random(): a seeded NumPy generator (`seed_random`).
"""
import ast
import inspect
import math as _math
import textwrap

import numpy as np

_py_float, _py_int, _py_range, _py_max, _py_min = float, int, range, max, min

DT = np.float32


def set_precision(name):
    global DT
    DT = {"f32": np.float32, "f64": np.float64}[name]


_rng = np.random.default_rng(0)


def seed_random(seed):
    global _rng
    _rng = np.random.default_rng(seed)


# ----------------------------------------------------------------------------- constants / no-ops
cuda = "cuda"
cpu = "cpu"
gpu = "gpu"
f32, f64, i32, i64 = np.float32, np.float64, np.int32, np.int64


def init(*args, **kwargs):
    pass


def template():
    return "template"


def static(x):
    return x


def _flt(x):
    return DT(x)


def _is_float(x):
    return isinstance(x, (_py_float, np.floating))


def _scalar(x):
    """What a field element / literal becomes inside a kernel: ints stay Python ints, floats take the working precision."""
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (_py_int, np.integer)):
        return _py_int(x)
    return DT(x)


# ----------------------------------------------------------------------------- Vector / Matrix values
class Vector:
    __array_ufunc__ = None

    def __init__(self, vals):
        self.v = [_scalar(x) for x in vals]

    def __len__(self):
        return len(self.v)

    def __iter__(self):
        return iter(self.v)

    def __getitem__(self, i):
        return self.v[i]

    def __setitem__(self, i, val):
        # a Taichi variable keeps the type it was created with
        self.v[i] = DT(val) if _is_float(self.v[i]) else _scalar(val)

    def _zip(self, o, op):
        if isinstance(o, Vector):
            assert len(o) == len(self)
            return Vector([op(a, b) for a, b in zip(self.v, o.v)])
        o = _scalar(o)
        return Vector([op(a, o) for a in self.v])

    def __add__(self, o):
        return self._zip(o, lambda a, b: a + b)

    __radd__ = __add__

    def __sub__(self, o):
        return self._zip(o, lambda a, b: a - b)

    def __rsub__(self, o):
        return self._zip(o, lambda a, b: b - a)

    def __mul__(self, o):
        return self._zip(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._zip(o, lambda a, b: a / b)

    def __neg__(self):
        return Vector([-a for a in self.v])

    def dot(self, o):
        acc = self.v[0] * o.v[0]
        for a, b in zip(self.v[1:], o.v[1:]):
            acc = acc + a * b
        return acc

    def norm(self):
        return np.sqrt(DT(self.dot(self)))

    def __repr__(self):
        return f"Vector({self.v})"


class Matrix:
    __array_ufunc__ = None

    def __init__(self, rows):
        self.m = [[_scalar(x) for x in r] for r in rows]

    def __getitem__(self, ij):
        i, j = ij
        return self.m[i][j]

    def __setitem__(self, ij, val):
        i, j = ij
        self.m[i][j] = _scalar(val)

    def transpose(self):
        return Matrix([list(c) for c in zip(*self.m)])

    def __matmul__(self, o):
        if isinstance(o, Vector):
            return Vector([Vector(r).dot(o) for r in self.m])
        cols = [Vector(c) for c in zip(*o.m)]
        return Matrix([[Vector(r).dot(c) for c in cols] for r in self.m])

    def __repr__(self):
        return f"Matrix({self.m})"


def sym_eig(A):
    """Stand-in (see the module docstring): eigenvalues ascending, eigenvectors in the columns."""
    a = np.array([[DT(x) for x in r] for r in A.m], dtype=DT)
    w, q = np.linalg.eigh(a)
    return Vector([DT(x) for x in w]), Matrix([[DT(x) for x in r] for r in q])


# ----------------------------------------------------------------------------- fields
class Field:
    def __init__(self, dtype, shape, n=None):
        self.is_int = dtype in (_py_int, np.int32, np.int64)
        self.n = n
        self.shape = (shape,) if isinstance(shape, (_py_int, np.integer)) else tuple(_py_int(s) for s in shape)
        full = self.shape + ((n,) if n is not None else ())
        self.a = np.zeros(full, dtype=np.int64 if self.is_int else DT)

    @staticmethod
    def _idx(idx):
        if isinstance(idx, Vector):
            return tuple(_py_int(x) for x in idx.v)
        if isinstance(idx, tuple):
            return tuple(_py_int(x) for x in idx)
        return (_py_int(idx),)

    def __getitem__(self, idx):
        x = self.a[self._idx(idx)]
        if self.n is not None:
            return Vector(list(x))
        return _py_int(x) if self.is_int else DT(x)

    def __setitem__(self, idx, val):
        if self.n is not None:
            self.a[self._idx(idx)] = [DT(x) for x in (val.v if isinstance(val, Vector) else val)]
        else:
            self.a[self._idx(idx)] = val

    def _atomic_add(self, idx, val):
        old = self[idx]
        self.a[self._idx(idx)] = old + (val if self.is_int else DT(val))
        return old

    def __iter__(self):      # struct-for: every index, C order
        if len(self.shape) == 1:
            return iter(_py_range(self.shape[0]))
        return iter(np.ndindex(*self.shape))

    def from_torch(self, t):
        self.a[...] = t.detach().cpu().numpy().reshape(self.a.shape)

    def from_numpy(self, arr):
        self.a[...] = np.asarray(arr).reshape(self.a.shape)

    def to_numpy(self):
        return self.a.astype(np.int32) if self.is_int else self.a.copy()

    def to_torch(self):
        import torch
        return torch.from_numpy(self.to_numpy())


FIELDS = []          # every field in creation order, so a driver can look at the reference's intermediate grids


def field(dtype, shape):
    f = Field(dtype, shape)
    FIELDS.append(f)
    return f


def _vector_field(n, dtype, shape):
    f = Field(dtype, shape, n)
    FIELDS.append(f)
    return f


Vector.field = staticmethod(_vector_field)


# ----------------------------------------------------------------------------- built-ins
def exp(x):
    return np.exp(DT(x))


def sqrt(x):
    return np.sqrt(DT(x))


def floor(x, dtype=None):
    r = _math.floor(x)
    return r if dtype is _py_int else DT(r)


def ceil(x, dtype=None):
    r = _math.ceil(x)
    return r if dtype is _py_int else DT(r)


def max(*args):  # noqa: A001
    out = args[0]
    for a in args[1:]:
        out = a if a > out else out
    return out if not any(_is_float(a) for a in args) else DT(out)


def min(*args):  # noqa: A001
    out = args[0]
    for a in args[1:]:
        out = a if a < out else out
    return out if not any(_is_float(a) for a in args) else DT(out)


def random(dtype=None):
    """Uniform in [0, 1), a float32 value in both precisions (so the two runs of a fixture place the same points)."""
    return DT(_py_min(np.float32(_rng.random()), np.float32(1.0 - 2.0 ** -24)))


def cast(x, dtype):
    return _py_int(x) if dtype in (_py_int, np.int32, np.int64) else DT(x)


class _Math:
    @staticmethod
    def mod(x, y):
        """taichi/math/mathimpl.py: x - y * floor(x / y)"""
        return x - y * _math.floor(x / y)


math = _Math()


# ----------------------------------------------------------------------------- kernel compilation
def _ti_range(*args):
    return _py_range(*(_py_int(a) for a in args))


class _Rewrite(ast.NodeTransformer):
    def visit_Constant(self, node):
        if isinstance(node.value, _py_float):
            return ast.copy_location(ast.Call(func=ast.Name(id="__ti_lit__", ctx=ast.Load()), args=[node], keywords=[]), node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if isinstance(f, ast.Name) and f.id == "range":
            node.func = ast.copy_location(ast.Name(id="__ti_range__", ctx=ast.Load()), f)
            return node
        if isinstance(f, ast.Attribute) and f.attr == "atomic_add" and isinstance(f.value, ast.Name) and f.value.id == "ti":
            target, val = node.args
            if isinstance(target, ast.Subscript):       # ti.atomic_add(field[idx], v) -> field._atomic_add(idx, v)
                new = ast.Call(func=ast.Attribute(value=target.value, attr="_atomic_add", ctx=ast.Load()), args=[target.slice, val], keywords=[])
                return ast.copy_location(new, node)
            assert isinstance(target, ast.Name)          # ti.atomic_add(name, v) -> ((old := name), (name := name + v), old)[2]
            load = lambda n: ast.Name(id=n, ctx=ast.Load())
            store = lambda n: ast.Name(id=n, ctx=ast.Store())
            tup = ast.Tuple(elts=[ast.NamedExpr(target=store("__ti_old__"), value=load(target.id)),
                                  ast.NamedExpr(target=store(target.id), value=ast.BinOp(left=load(target.id), op=ast.Add(), right=val)),
                                  load("__ti_old__")], ctx=ast.Load())
            return ast.copy_location(ast.Subscript(value=tup, slice=ast.Constant(value=2), ctx=ast.Load()), node)
        return node


def _recompile(fn):
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    fdef = tree.body[0]
    assert isinstance(fdef, ast.FunctionDef) and fdef.name == fn.__name__
    annotations = [(a.arg, ast.unparse(a.annotation) if a.annotation is not None else None) for a in fdef.args.args]
    fdef.decorator_list = []
    for a in fdef.args.args:
        a.annotation = None
    fdef.returns = None
    tree = ast.fix_missing_locations(_Rewrite().visit(tree))
    ast.increment_lineno(tree, fn.__code__.co_firstlineno - 1)          # tracebacks point at the reference's own lines
    code = compile(tree, inspect.getsourcefile(fn), "exec")
    glb = fn.__globals__                                                  # live: functions defined later resolve at call time
    glb["__ti_lit__"] = _flt
    glb["__ti_range__"] = _ti_range
    scratch = {}
    exec(code, glb, scratch)
    return scratch[fn.__name__], annotations


def func(fn):
    return _recompile(fn)[0]


KERNEL_LOG = []       # names of the kernels in launch order
KERNEL_HOOK = None    # optional callable(name, {argument name: value}) run after every kernel


def kernel(fn):
    body, annotations = _recompile(fn)
    names = [n for n, _ in annotations]

    def launch(*args, **kwargs):
        bound = dict(zip(names, args))
        bound.update(kwargs)
        for name, ann in annotations:
            if ann == "float":
                bound[name] = DT(np.float32(bound[name]))      # an f32 kernel argument; the float64 mode computes on the same value
            elif ann == "int":
                bound[name] = _py_int(bound[name])
        out = body(**bound)
        KERNEL_LOG.append(fn.__name__)
        if KERNEL_HOOK is not None:
            KERNEL_HOOK(fn.__name__, bound)
        return out
    launch.__name__ = fn.__name__
    return launch

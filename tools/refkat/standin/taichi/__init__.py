"""A stand-in for the `taichi` package (0.7.x API surface the reference uses), BUILD CONTAINER ONLY.

Taichi is not installable here.  This module lets the reference's own source files under /root/reference be
IMPORTED AND EXECUTED as plain Python by tools/refkat/make_refkat.py: `@ti.func` / `@ti.kernel` / `@ti.data_oriented` are identity
decorators, `ti.Vector` is a small fp32 vector whose every arithmetic operation rounds to fp32 (as Taichi's f32 code does, one
IEEE operation per source-level operation), fields are numpy arrays, `ti.random()` pops from a queue the caller fills.  What this
gives is a TRANSCRIPTION CHECK -- the reference's source text decides every formula, constant, branch and operand order of the golden
vectors -- not a run of the reference: Taichi's code generator (fast-math, its own sin / cos / pow) is not reproduced.
Transcendental functions are evaluated in float64 and rounded once to fp32 unless `set_math()` installs other ones."""
import math as _math
import numpy as np

f32 = np.float32
f64 = np.float64
i32 = np.int32
u32 = np.uint32
i, j, k, ij, ijk = "i", "j", "k", "ij", "ijk"
cpu = "cpu"; gpu = "gpu"


def _s(x):
    """scalar of a ti.func: python float -> fp32, python int stays an int"""
    if isinstance(x, (float, np.floating)):
        return np.float32(x)
    if isinstance(x, (bool, np.bool_)):
        return int(x)
    return x


def init(*a, **kw):
    pass


def _copy(x):
    return x.copy() if isinstance(x, (Vector, Matrix)) else x


def _taichi_scope(f):
    """What the decorators do here: in Taichi scope `a = b` COPIES a vector (min_v3 = v1; max_v3 = v1; min_v3[k] = ... in accel/LBvh.py:404-411
    must not touch max_v3), in Python it makes a second name for one object.  The function is recompiled from its source with every plain
    `name = name` assignment turned into `name = copy(name)`; nothing else changes (same file name and line numbers, same globals)."""
    import ast, inspect, textwrap
    try:
        src = textwrap.dedent(inspect.getsource(f))
    except (OSError, TypeError):
        return f
    tree = ast.parse(src)
    fd = tree.body[0]
    fd.decorator_list = []

    class T(ast.NodeTransformer):
        def visit_Assign(self, node):
            self.generic_visit(node)
            if isinstance(node.value, ast.Name) and all(isinstance(t, ast.Name) for t in node.targets):
                node.value = ast.Call(func=ast.Name(id="__ti_copy__", ctx=ast.Load()), args=[node.value], keywords=[])
            return node
    T().visit(tree)
    ast.increment_lineno(tree, f.__code__.co_firstlineno - 1)          # line k of the snippet is line co_firstlineno + k - 1 of the file
    ast.fix_missing_locations(tree)
    g = f.__globals__
    g["__ti_copy__"] = _copy
    ns = {}
    exec(compile(tree, inspect.getsourcefile(f) or "<taichi-scope>", "exec"), g, ns)
    return ns[f.__name__]


def func(f): return _taichi_scope(f)
def kernel(f): return _taichi_scope(f)
def pyfunc(f): return f
def data_oriented(c): return c
def static(*x): return x[0] if len(x) == 1 else x
def template(): return None


# ---- random numbers: the caller decides what ti.random() returns ----------------------------------
_rand_source = None


def set_random(source):
    """source: a callable returning the next number, or a list that is popped from the front"""
    global _rand_source
    _rand_source = source


def random(dtype=None):
    if callable(_rand_source):
        return np.float32(_rand_source())
    return np.float32(_rand_source.pop(0))


# ---- math: one rounding to fp32 per call ------------------------------------------------------------
_math_impl = {}


def set_math(table):
    """table: {'sin': f, ...} of fp32 -> fp32 callables (make_refkat.py can install the oracle's own polynomials so that a whole
    render agrees to the last bit where the transcription is right)"""
    _math_impl.update(table)


def _m1(name, ref):
    def f(x):
        if isinstance(x, Vector):
            return Vector([f(e) for e in x.e])
        if name in _math_impl:
            return np.float32(_math_impl[name](np.float32(x)))
        with np.errstate(all="ignore"):
            return np.float32(ref(np.float64(np.float32(x))))
    return f


sqrt = _m1("sqrt", np.sqrt)
sin = _m1("sin", np.sin)
cos = _m1("cos", np.cos)
tan = _m1("tan", np.tan)
exp = _m1("exp", np.exp)
log = _m1("log", np.log)
acos = _m1("acos", np.arccos)
asin = _m1("asin", np.arcsin)
floor = _m1("floor", np.floor)
ceil = _m1("ceil", np.ceil)


def atan2(y, x):
    if "atan2" in _math_impl:
        return np.float32(_math_impl["atan2"](np.float32(y), np.float32(x)))
    return np.float32(np.arctan2(np.float64(np.float32(y)), np.float64(np.float32(x))))


def pow_(x, y):
    """what the reference's `pow(a, b)` means inside a ti.func (make_refkat.py binds the name `pow` of the reference modules to it)"""
    if "pow" in _math_impl:
        return np.float32(_math_impl["pow"](np.float32(x), np.float32(y)))
    with np.errstate(all="ignore"):
        return np.float32(np.power(np.float64(np.float32(x)), np.float64(np.float32(y))))


def cast(x, dtype):
    if isinstance(x, Vector):
        return Vector([cast(e, dtype) for e in x.e])
    if dtype in (i32, int):
        return int(np.float32(x)) if isinstance(x, (float, np.floating)) else int(x)      # truncation, as Taichi's f32 -> i32 cast
    if dtype is u32:
        return int(x) & 0xffffffff
    return np.float32(x)


def bit_cast(x, dtype):
    if dtype is i32:
        return int(np.array([x], np.float32).view(np.int32)[0])
    if dtype is f32:
        v = int(x)
        v = (v + (1 << 31)) % (1 << 32) - (1 << 31)
        return np.array([v], np.int32).view(np.float32)[0]
    raise TypeError(dtype)


class _IntRef(int):
    """an element of a scalar integer field: an int that remembers where it lives, so that ti.atomic_add(field[i], v) can write"""
    def __new__(cls, v, arr, idx):
        o = int.__new__(cls, v); o._arr, o._idx = arr, idx
        return o


def atomic_add(a, b):
    old = int(a)
    a._arr[a._idx] = old + int(b)
    return old


def vmin(a, b):
    """what `min(a, b)` means inside a ti.func (elementwise on vectors); make_refkat.py binds the reference modules' `min` / `max` to these"""
    if isinstance(a, Vector) or isinstance(b, Vector):
        n = len(a) if isinstance(a, Vector) else len(b)
        return Vector([vmin(a[k] if isinstance(a, Vector) else a, b[k] if isinstance(b, Vector) else b) for k in range(n)])
    return b if b < a else a


def vmax(a, b):
    if isinstance(a, Vector) or isinstance(b, Vector):
        n = len(a) if isinstance(a, Vector) else len(b)
        return Vector([vmax(a[k] if isinstance(a, Vector) else a, b[k] if isinstance(b, Vector) else b) for k in range(n)])
    return b if b > a else a


# ---- vectors and matrices -----------------------------------------------------------------------------
class Vector:
    """ti.Vector of fp32 (or integer) components; every binary operation is done component by component in fp32."""
    __array_priority__ = 1000

    def __init__(self, e, dt=None):
        self.e = [_s(x) for x in (e.e if isinstance(e, Vector) else e)]
        if dt is f32:
            self.e = [np.float32(x) for x in self.e]

    # construction helpers used by the reference
    @staticmethod
    def field(n, dtype=None, shape=None, **kw):
        return Field(dtype, shape, n)

    @staticmethod
    def zero(dt, n):
        return Vector([0.0] * n)

    n = property(lambda s: len(s.e))

    def __len__(self): return len(self.e)
    def __iter__(self): return iter(self.e)
    def __getitem__(self, k): return self.e[k]
    def __setitem__(self, k, v): self.e[k] = _s(v)
    x = property(lambda s: s.e[0], lambda s, v: s.__setitem__(0, v))
    y = property(lambda s: s.e[1], lambda s, v: s.__setitem__(1, v))
    z = property(lambda s: s.e[2], lambda s, v: s.__setitem__(2, v))
    w = property(lambda s: s.e[3], lambda s, v: s.__setitem__(3, v))

    def _bin(self, o, op, rev=False):
        if isinstance(o, Vector):
            assert len(o.e) == len(self.e)
            pairs = zip(self.e, o.e)
        else:
            pairs = ((a, _s(o)) for a in self.e)
        with np.errstate(all="ignore"):
            return Vector([op(b, a) if rev else op(a, b) for a, b in pairs])

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: a / b, True)
    def __neg__(self): return Vector([-a for a in self.e])
    def __repr__(self): return "Vector(%s)" % (self.e,)

    def dot(self, o):
        acc = self.e[0] * o.e[0]
        for a, b in zip(self.e[1:], o.e[1:]):
            acc = acc + a * b                       # Taichi unrolls the sum left to right
        return acc

    def cross(self, o):
        a, b = self.e, o.e
        return Vector([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])

    def sum(self):
        acc = self.e[0]
        for a in self.e[1:]:
            acc = acc + a                           # left to right, as Taichi unrolls it
        return acc

    def norm_sqr(self): return self.dot(self)
    def norm(self, eps=0): return sqrt(self.norm_sqr() + eps) if eps else sqrt(self.norm_sqr())

    def normalized(self, eps=0):
        # taichi 0.7 matrix.py: invlen = 1 / (self.norm() + eps); return invlen * self
        with np.errstate(all="ignore"):
            invlen = np.float32(1.0) / (self.norm() + np.float32(eps))
        return invlen * self

    def to_numpy(self): return np.array(self.e)
    def copy(self): return Vector(list(self.e))


class Matrix:
    def __init__(self, rows, dt=None):
        self.m = [[_s(x) for x in r] for r in rows]

    @staticmethod
    def field(n, m, dtype=None, shape=None, **kw):
        return Field(dtype, shape, (n, m))

    @staticmethod
    def rows(rs):
        return Matrix([list(r.e) for r in rs])

    def __getitem__(self, ij_): return self.m[ij_[0]][ij_[1]]

    def __matmul__(self, v):
        out = []
        for r in self.m:
            acc = r[0] * v.e[0]
            for a, b in zip(r[1:], v.e[1:]):
                acc = acc + a * b
            out.append(acc)
        return Vector(out)

    def inverse(self):
        return Matrix(np.linalg.inv(np.array(self.m, np.float64)).astype(np.float32).tolist())

    def transpose(self):
        return Matrix([list(c) for c in zip(*self.m)])

    def copy(self):
        return Matrix([list(r) for r in self.m])


# ---- fields --------------------------------------------------------------------------------------------
class _Place:
    def __init__(self, shape): self.shape = shape
    def place(self, *fields):
        for f in fields:
            f._alloc(self.shape)


class _Root:
    def dense(self, axes, shape):
        return _Place(tuple(shape) if isinstance(shape, (list, tuple)) else (int(shape),))


root = _Root()


def _pot(n):
    p = 1
    while p < n:
        p <<= 1
    return p


class Field:
    """ti.field / ti.Vector.field / ti.Matrix.field backed by a numpy array; element access returns python-side Vector / Matrix copies.
    As in Taichi's dense SNodes, every axis is padded to a power of two and an index is taken modulo that size (bit extraction): the
    reference writes to index -1 in places (BDPT_RGB.py:472-477 restores `light[l-1]` / `eye[e-2]` with l = 0 / e = 1), which in Taichi
    lands in the padding behind a depth axis of 6 or 7 -- not, as a Python index would, on the last real element."""
    def __init__(self, dtype, shape=None, inner=None):
        self.dtype = np.int32 if dtype in (i32, int) else (np.uint32 if dtype is u32 else np.float32)
        self.inner = inner
        self.a = None
        if shape is not None:
            self._alloc(tuple(shape) if isinstance(shape, (list, tuple)) else (int(shape),))

    def _alloc(self, shape):
        inner = () if self.inner is None else ((self.inner,) if isinstance(self.inner, int) else tuple(self.inner))
        self.shape = tuple(int(x) for x in shape)
        self.pot = tuple(_pot(x) for x in self.shape)
        self.a = np.zeros(self.pot + inner, self.dtype)

    def _logical(self):
        return self.a[tuple(slice(0, n) for n in self.shape)]

    def from_numpy(self, arr):
        arr = np.asarray(arr)
        nin = 0 if self.inner is None else (1 if isinstance(self.inner, int) else 2)
        shape = arr.shape[: arr.ndim - nin]
        if self.a is None or self.shape != tuple(shape):
            self._alloc(shape)
        self._logical()[...] = arr.astype(self.dtype)

    def to_numpy(self): return self._logical().copy()

    def _idx(self, k):
        if isinstance(k, Vector):                    # field[ti.Vector([u, v])]
            k = tuple(k.e)
        k = k if isinstance(k, tuple) else (k,)
        return tuple(int(x) & (p - 1) for x, p in zip(k, self.pot))

    def __getitem__(self, k):
        v = self.a[self._idx(k)]
        if self.inner is None:
            return _IntRef(int(v), self.a, self._idx(k)) if self.dtype != np.float32 else np.float32(v)
        if isinstance(self.inner, int):
            return _Row(self.a, self._idx(k))
        return Matrix(v.tolist())

    def __setitem__(self, k, v):
        if isinstance(v, Vector):
            self.a[self._idx(k)] = np.array([x for x in v.e], self.dtype)
        elif isinstance(v, Matrix):
            self.a[self._idx(k)] = np.array(v.m, self.dtype)
        else:
            self.a[self._idx(k)] = v

    def __iter__(self):
        if len(self.shape) == 1:                     # `for i in field` of a 1-D field: scalar indices
            return iter(range(self.shape[0]))
        return iter(np.ndindex(*self.shape))


class _Row(Vector):
    """a vector element of a field: reads like a Vector, component writes go through to the field (material[index][5] = ...)"""
    def __init__(self, arr, idx):
        self._arr, self._idx2 = arr, idx
        row = arr[idx]
        self.e = [int(x) for x in row] if arr.dtype != np.float32 else [np.float32(x) for x in row]

    def __setitem__(self, k, v):
        self._arr[self._idx2 + (k,)] = v
        self.e[k] = _s(v)


def field(dtype=None, shape=None, **kw):
    return Field(dtype, shape)


class GUI:
    def __init__(self, *a, **kw): pass

"""Stand-in for taichi_glsl 0.0.9 (the functions the reference calls), see ../taichi/__init__.py.  BUILD CONTAINER ONLY."""
import numpy as np
import taichi as ti
from taichi import Vector


def _map(f, *a):
    if any(isinstance(x, Vector) for x in a):
        n = max(len(x) for x in a if isinstance(x, Vector))
        return Vector([f(*[(x[k] if isinstance(x, Vector) else x) for x in a]) for k in range(n)])
    return f(*a)


def _f(x): return np.float32(x) if isinstance(x, (float, np.floating)) else x


def reflect(I, N): return I - 2 * N.dot(I) * N                       # taichi_glsl/vector.py
def mix(x, y, a): return _map(lambda p, q, r: _f(p) * (1 - _f(r)) + _f(q) * _f(r), x, y, a)      # x * (1 - a) + y * a
def clamp(x, xmin=0, xmax=1): return _map(lambda p, lo, hi: min(_f(hi), max(_f(lo), _f(p))), x, xmin, xmax)
def sign(x, edge=0): return _map(lambda p: np.float32((p > edge) * 1.0 - (p < edge) * 1.0), x)
def fract(x): return _map(lambda p: _f(p) - ti.floor(p), x)
def length(x): return x.norm()
def normalize(x): return x.normalized()
def dot(a, b): return a.dot(b)
def cross(a, b): return a.cross(b)
def atan(y, x=None): return ti.atan2(y, x) if x is not None else ti.atan2(y, 1.0)
def acos(x): return ti.acos(x)
def asin(x): return ti.asin(x)
def sin(x): return ti.sin(x)
def cos(x): return ti.cos(x)
def tan(x): return ti.tan(x)
def sqrt(x): return ti.sqrt(x)
def floor(x): return ti.floor(x)
def vec2(*a): return Vector(list(a))
def vec3(*a): return Vector(list(a) if len(a) == 3 else [a[0]] * 3)
def vec4(*a): return Vector(list(a))

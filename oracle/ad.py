"""oracle/ad.py — TEST INFRASTRUCTURE ONLY (never imported by the product path stark_amd/).

Vectorised second-order forward-mode automatic differentiation in numpy.

The reference obtains gradients and Hessians of every energy by symbolic differentiation of the energy expression
(symx/src/symbol/diff.cpp:7-83, driven from symx/src/solver/second_order/SecondOrderCompiledPotential.cpp:64-69).
The oracle restates the *energy expressions* (oracle/energies.py) and differentiates them with this independent,
numerically exact (to round-off) AD: a value `v[E]`, gradient `g[E,N]` and Hessian `h[E,N,N]` per element, where N is
the number of element DoFs. Nothing here is performance relevant.
"""
from __future__ import annotations

import numpy as np


class D2:
    """Batch of scalars with first and second derivatives w.r.t. N independent variables."""

    __slots__ = ("v", "g", "h")
    __array_priority__ = 1000

    def __init__(self, v, g, h):
        self.v = v
        self.g = g
        self.h = h

    # ---- construction ------------------------------------------------------------------------------------------
    @staticmethod
    def const(v, n):
        v = np.asarray(v, dtype=np.float64)
        return D2(v, np.zeros(v.shape + (n,)), np.zeros(v.shape + (n, n)))

    @staticmethod
    def var(v, i, n):
        v = np.asarray(v, dtype=np.float64)
        g = np.zeros(v.shape + (n,))
        g[..., i] = 1.0
        return D2(v, g, np.zeros(v.shape + (n, n)))

    @property
    def n(self):
        return self.g.shape[-1]

    def _lift(self, o):
        if isinstance(o, D2):
            return o
        return D2.const(np.broadcast_to(np.asarray(o, dtype=np.float64), self.v.shape), self.n)

    # ---- generic unary chain rule: y = f(x), with f' and f'' given as arrays --------------------------------------
    def _chain(self, f, df, ddf):
        g = df[..., None] * self.g
        h = df[..., None, None] * self.h + ddf[..., None, None] * (self.g[..., :, None] * self.g[..., None, :])
        return D2(f, g, h)

    # ---- arithmetic ------------------------------------------------------------------------------------------------
    def __add__(self, o):
        if isinstance(o, D2):
            return D2(self.v + o.v, self.g + o.g, self.h + o.h)
        return D2(self.v + o, self.g, self.h)

    __radd__ = __add__

    def __neg__(self):
        return D2(-self.v, -self.g, -self.h)

    def __sub__(self, o):
        if isinstance(o, D2):
            return D2(self.v - o.v, self.g - o.g, self.h - o.h)
        return D2(self.v - o, self.g, self.h)

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        if isinstance(o, D2):
            v = self.v * o.v
            g = self.v[..., None] * o.g + o.v[..., None] * self.g
            cross = self.g[..., :, None] * o.g[..., None, :]
            h = self.v[..., None, None] * o.h + o.v[..., None, None] * self.h + cross + np.swapaxes(cross, -1, -2)
            return D2(v, g, h)
        o = np.asarray(o, dtype=np.float64)
        return D2(self.v * o, self.g * o[..., None], self.h * o[..., None, None])

    __rmul__ = __mul__

    def inv(self):
        r = 1.0 / self.v
        return self._chain(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o):
        if isinstance(o, D2):
            return self * o.inv()
        return self * (1.0 / np.asarray(o, dtype=np.float64))

    def __rtruediv__(self, o):
        return self.inv() * o

    def powN(self, k: int):
        v = self.v
        return self._chain(v ** k, k * v ** (k - 1), k * (k - 1) * v ** (k - 2) if k >= 2 else np.zeros_like(v))

    def sqrt(self):
        s = np.sqrt(self.v)
        with np.errstate(divide="ignore", invalid="ignore"):
            return self._chain(s, 0.5 / s, -0.25 / (s * self.v))

    def log(self):
        r = 1.0 / self.v
        return self._chain(np.log(self.v), r, -r * r)

    def acos(self):
        x = self.v
        s = 1.0 / np.sqrt(1.0 - x * x)
        return self._chain(np.arccos(x), -s, -x * s * s * s)

    def atan(self):
        x = self.v
        d = 1.0 / (1.0 + x * x)
        return self._chain(np.arctan(x), d, -2.0 * x * d * d)

    def cos(self):
        return self._chain(np.cos(self.v), -np.sin(self.v), -np.cos(self.v))

    def sin(self):
        return self._chain(np.sin(self.v), np.cos(self.v), -np.sin(self.v))


def where(cond, a, b):
    """Reference `branch(cond, a, b)` (symx Scalar branch): per-element selection of value AND derivatives."""
    if not isinstance(a, D2) and not isinstance(b, D2):
        return np.where(cond, a, b)
    ref = a if isinstance(a, D2) else b
    a = ref._lift(a)
    b = ref._lift(b)
    c = np.asarray(cond)
    # derivatives of the unselected branch may be nan/inf (e.g. sqrt at 0); np.where discards them
    return D2(np.where(c, a.v, b.v), np.where(c[..., None], a.g, b.g), np.where(c[..., None, None], a.h, b.h))


# ---- small vector / matrix helpers on lists of D2 (or plain arrays) ---------------------------------------------------
def dot(a, b):
    s = a[0] * b[0]
    for i in range(1, len(a)):
        s = s + a[i] * b[i]
    return s


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def sub(a, b):
    return [x - y for x, y in zip(a, b)]


def add(a, b):
    return [x + y for x, y in zip(a, b)]


def scale(s, a):
    return [s * x for x in a]


def sqnorm(a):
    return dot(a, a)


def norm(a):
    n = sqnorm(a)
    return n.sqrt() if isinstance(n, D2) else np.sqrt(n)


def normalized(a):
    n = norm(a)
    return [x / n for x in a]


def matmul(A, B):
    """A: r x k, B: k x c as nested lists."""
    r, k, c = len(A), len(B), len(B[0])
    return [[dot([A[i][m] for m in range(k)], [B[m][j] for m in range(k)]) for j in range(c)] for i in range(r)]


def transpose(A):
    return [[A[i][j] for i in range(len(A))] for j in range(len(A[0]))]


def det3(A):
    return (A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1])
            - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
            + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]))


def inv3(A):
    d = det3(A)
    c = [[A[1][1] * A[2][2] - A[1][2] * A[2][1], A[0][2] * A[2][1] - A[0][1] * A[2][2], A[0][1] * A[1][2] - A[0][2] * A[1][1]],
         [A[1][2] * A[2][0] - A[1][0] * A[2][2], A[0][0] * A[2][2] - A[0][2] * A[2][0], A[0][2] * A[1][0] - A[0][0] * A[1][2]],
         [A[1][0] * A[2][1] - A[1][1] * A[2][0], A[0][1] * A[2][0] - A[0][0] * A[2][1], A[0][0] * A[1][1] - A[0][1] * A[1][0]]]
    return [[c[i][j] / d for j in range(3)] for i in range(3)]


def inv2(A):
    d = A[0][0] * A[1][1] - A[0][1] * A[1][0]
    return [[A[1][1] / d, -1.0 * A[0][1] / d], [-1.0 * A[1][0] / d, A[0][0] / d]]


def frob_sq(A):
    s = None
    for row in A:
        for x in row:
            s = x * x if s is None else s + x * x
    return s


def trace(A):
    s = A[0][0]
    for i in range(1, len(A)):
        s = s + A[i][i]
    return s

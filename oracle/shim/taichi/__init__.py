"""Serial fake-`taichi` used ONLY to generate golden vectors (tests/golden/make_golden.py).

TEST INFRASTRUCTURE - never imported by the product package.

The reference (takah29/2d-fluid-simulator) writes all of its arithmetic as Taichi-DSL Python
(`@ti.kernel` / `@ti.func` bodies in fs/*.py).  Taichi 1.7.4 itself is not installable in this
image (needs Python >= 3.13, no wheel, no network), so the reference's *own kernel source* is
executed here under this module, which gives the DSL a deterministic serial meaning:

  * decorators are identity (float/int annotated args are cast, as Taichi casts kernel args),
  * fields are NumPy arrays, scalars are np.float32 (or np.float64 with default_fp=f64),
  * `for i, j in field` iterates serially in (i-major, j-minor) order - the order a
    single-threaded Taichi CPU run visits a dense field (hazard H1),
  * Python-float constants stay Python floats until they meet a field value, so constant
    sub-expressions fold in f64 and are rounded once, as Taichi's AST pass does (H6),
  * ti.min / ti.max are NaN-ignoring (llvm.minnum/maxnum-like; H4),
  * out-of-bounds reads follow OOB_POLICY ("clamp" or "zero"; H2/H3) and are logged,
  * matrix @ vector is an explicit left-to-right f32 multiply-add chain (no BLAS / FMA).

Only the API subset the reference uses is provided.
"""
import builtins
import ast
import inspect
import textwrap
import itertools

import numpy as np

_bmin, _bmax = builtins.min, builtins.max

f32 = np.float32
f64 = np.float64
i32 = np.int32
u8 = np.uint8
cpu = "cpu"
gpu = "gpu"

_FP = np.float32          # default float type of fields / scalars
OOB_POLICY = "clamp"      # "clamp" | "zero"
OOB_LOG = {}


def init(arch=None, default_fp=None, **_kw):
    global _FP
    if default_fp is not None:
        _FP = default_fp


def set_default_fp(fp):
    """f64 'truth' runs: every field/scalar the reference declares as ti.f32 becomes fp."""
    global _FP, f32
    _FP = fp
    f32 = fp


def template():
    return "template"


def static(x):
    return x


class Vec(np.ndarray):
    """Small vector value; remembers the field element it was read from so that
    `field[i, j].x = value` (boundary_condition.py:39) writes through."""

    _owner = None
    _idx = None

    def __new__(cls, data):
        a = np.asarray(data)
        if a.dtype.kind == "f" or a.dtype == object:
            a = a.astype(_FP)
        return a.view(cls)

    def __array_finalize__(self, obj):
        self._owner = None
        self._idx = None

    def _setc(self, k, v):
        np.ndarray.__setitem__(self, k, v)
        if self._owner is not None:
            self._owner.arr[self._idx][k] = v

    x = property(lambda s: np.ndarray.__getitem__(s, 0), lambda s, v: s._setc(0, v))
    y = property(lambda s: np.ndarray.__getitem__(s, 1), lambda s, v: s._setc(1, v))
    z = property(lambda s: np.ndarray.__getitem__(s, 2), lambda s, v: s._setc(2, v))

    def __matmul__(self, o):
        # (n x k) @ (k,) as Taichi unrolls it: ((m0*v0 + m1*v1) + m2*v2) + ...
        m = np.asarray(self)
        w = np.asarray(o).astype(m.dtype)
        acc = m[..., 0] * w[0]
        for k in range(1, w.shape[0]):
            acc = acc + m[..., k] * w[k]
        return np.asarray(acc).view(Vec)

    def norm(self):
        a = np.asarray(self)
        s = a[0] * a[0]
        for k in range(1, a.shape[0]):
            s = s + a[k] * a[k]
        return np.sqrt(s)


class _VectorNS:
    def __or__(self, o):
        return self

    def __ror__(self, o):
        return self

    def __call__(self, data):
        return Vec(list(data))

    @staticmethod
    def field(n, dtype, shape):
        return Field(np.zeros(tuple(shape) + (n,), dtype=dtype), vec=True)


Vector = _VectorNS()


class _MatrixNS:
    @staticmethod
    def cols(cols):
        return np.stack([np.asarray(c) for c in cols], axis=-1).view(Vec)


Matrix = _MatrixNS()


def field(dtype, shape):
    return Field(np.zeros(tuple(shape), dtype=dtype), vec=False)


class Field:
    def __init__(self, arr, vec):
        self.arr = arr
        self.vec = vec
        self.shape = arr.shape[:2]

    def _resolve(self, idx, write):
        if isinstance(idx, np.ndarray):
            idx = (int(idx[0]), int(idx[1]))
        i, j = int(idx[0]), int(idx[1])
        if 0 <= i < self.shape[0] and 0 <= j < self.shape[1]:
            return (i, j)
        fr = inspect.stack()[2]
        key = (fr.function, fr.lineno, "W" if write else "R")
        OOB_LOG[key] = OOB_LOG.get(key, 0) + 1
        if write or OOB_POLICY == "zero":
            return None
        return (_bmin(_bmax(i, 0), self.shape[0] - 1), _bmin(_bmax(j, 0), self.shape[1] - 1))

    def __getitem__(self, idx):
        ij = self._resolve(idx, False)
        if ij is None:
            if self.vec:
                return Vec(np.zeros(self.arr.shape[2], self.arr.dtype))
            return self.arr.dtype.type(0)
        if self.vec:
            v = Vec(self.arr[ij].copy())
            v._owner = self
            v._idx = ij
            return v
        return self.arr[ij]

    def __setitem__(self, idx, val):
        ij = self._resolve(idx, True)
        if ij is not None:
            self.arr[ij] = val

    def __iter__(self):
        return iter(itertools.product(range(self.shape[0]), range(self.shape[1])))

    def from_numpy(self, a):
        self.arr[...] = a

    def to_numpy(self):
        return self.arr.copy()

    def fill(self, v):
        self.arr[...] = v


def _cast_args(fn):
    anns = [p.annotation for p in inspect.signature(fn).parameters.values()]

    def wrapper(*args, **kw):
        args = list(args)
        for k, a in enumerate(args):
            if k < len(anns):
                if anns[k] is float or anns[k] is f32 or anns[k] is np.float32:
                    args[k] = _FP(a)
                elif anns[k] is int:
                    args[k] = int(a)
        return fn(*args, **kw)

    wrapper.__wrapped__ = fn
    return wrapper


def _ipow(a, n):
    """Taichi lowers `x ** <integer literal>` to multiplications (demote_operations: exponentiation by squaring), so a run-time
    f32 value squared is x * x rounded once - NOT libm's powf(x, 2), which numpy uses for scalar ** and which is not correctly
    rounded (a 1-ulp difference roughly once per 10^5 values; found by tests/golden/fuzz_oracle_vs_reference.py).  Python floats are
    compile-time constants that Taichi folds in double precision: they keep Python's own power."""
    if isinstance(a, (float, int)) and not isinstance(a, np.generic):
        return a ** n
    result, base = None, a
    while n:
        if n & 1:
            result = base if result is None else result * base
        n >>= 1
        if n:
            base = base * base
    return result


class _DemoteIntPow(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if (isinstance(node.op, ast.Pow) and isinstance(node.right, ast.Constant) and type(node.right.value) is int
                and 1 <= node.right.value <= 16):
            return ast.copy_location(ast.Call(func=ast.Name(id="__ti_ipow", ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
        return node


def _demote_int_pow(fn):
    """Recompile fn with every `expr ** <int literal>` replaced by __ti_ipow(expr, n)."""
    if fn.__closure__:
        return fn
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    except (OSError, TypeError, SyntaxError):
        return fn
    fdef = tree.body[0]
    if not isinstance(fdef, ast.FunctionDef) or "**" not in ast.unparse(fdef):
        return fn
    fdef.decorator_list = []
    tree = ast.fix_missing_locations(_DemoteIntPow().visit(tree))
    ns = {}
    glb = fn.__globals__
    glb.setdefault("__ti_ipow", _ipow)
    exec(compile(tree, inspect.getsourcefile(fn) or "<shim>", "exec"), glb, ns)
    new = ns[fdef.name]
    new.__defaults__, new.__kwdefaults__ = fn.__defaults__, fn.__kwdefaults__
    new.__qualname__, new.__module__, new.__doc__ = fn.__qualname__, fn.__module__, fn.__doc__
    return new


def func(fn):
    return _cast_args(_demote_int_pow(fn))


def kernel(fn):
    return _cast_args(_demote_int_pow(fn))


def data_oriented(cls):
    return cls


def _wrap(x):
    return x.view(Vec) if isinstance(x, np.ndarray) and x.ndim else x


def max(a, b):  # noqa: A001
    return _wrap(np.fmax(a, b))


def min(a, b):  # noqa: A001
    return _wrap(np.fmin(a, b))


def abs(a):  # noqa: A001
    return np.abs(a)


def sqrt(a):
    return np.sqrt(_FP(a))


def floor(a):
    return np.floor(a)


def atan2(a, b):
    return np.arctan2(a, b)

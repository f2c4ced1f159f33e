"""A small SPIR-V interpreter, enough to execute the reference's COMPILED compute shaders
(shaders/*.comp.spv) one invocation at a time on the CPU.

TEST INFRASTRUCTURE ONLY.  It exists so that golden vectors can be produced from the reference's own
artefact -- the SPIR-V binaries it ships -- instead of from a restatement of its GLSL sources: the
binaries are read from /root/reference when tests/golden/make_spv_golden.py is run, and only the
resulting vectors are committed.  Nothing of the reference is embedded here: this file implements the
(public) SPIR-V / GLSL.std.450 semantics of the ~60 opcodes those modules use.

Numerics: every float operation is evaluated in IEEE binary32 with one rounding per SPIR-V instruction
(numpy float32 scalars), i.e. without contraction; transcendental extended instructions use numpy's
float32 functions.  A real driver may fuse or approximate -- which is exactly the latitude the parity
tolerances leave -- but integer results that only depend on + - * and comparisons (escape indices) are
exact by construction.
"""
from __future__ import annotations

import struct
from fractions import Fraction

import numpy as np

F32 = np.float32


class Cell:
    """One OpVariable: a mutable slot holding a (possibly nested list) value."""
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value


class Ptr:
    __slots__ = ("cell", "path")

    def __init__(self, cell, path=()):
        self.cell, self.path = cell, path

    def load(self):
        v = self.cell.value
        for i in self.path:
            v = v[i]
        return _copy(v)

    def store(self, x):
        x = _copy(x)
        if not self.path:
            self.cell.value = x
            return
        v = self.cell.value
        for i in self.path[:-1]:
            v = v[i]
        v[self.path[-1]] = x


def _copy(v):
    return [_copy(e) for e in v] if isinstance(v, list) else v


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _lift1(fn):
    def g(a):
        return [g(e) for e in a] if isinstance(a, list) else fn(a)
    return g


def _lift2(fn):
    def g(a, b):
        if isinstance(a, list):
            return [g(x, b[k] if isinstance(b, list) else b) for k, x in enumerate(a)]
        if isinstance(b, list):
            return [g(a, y) for y in b]
        return fn(a, b)
    return g


def _lift3(fn):
    def g(a, b, c):
        if isinstance(a, list) or isinstance(b, list) or isinstance(c, list):
            n = len(next(v for v in (a, b, c) if isinstance(v, list)))
            pick = lambda v, k: v[k] if isinstance(v, list) else v  # noqa: E731
            return [g(pick(a, k), pick(b, k), pick(c, k)) for k in range(n)]
        return fn(a, b, c)
    return g


def _fma32(a, b, c):
    """Correctly rounded binary32 fma (exact rational arithmetic, then one rounding)."""
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        return F32(np.float64(a) * np.float64(b) + np.float64(c))
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    return _round_fraction_f32(exact)


def _round_fraction_f32(q: Fraction):
    if q == 0:
        return F32(0.0)
    d = float(q)                       # nearest double; then fix the double rounding with a comparison
    f = F32(d)
    lo, hi = np.nextafter(f, F32(-np.inf)), np.nextafter(f, F32(np.inf))
    best = min((f, lo, hi), key=lambda c: (abs(Fraction(float(c)) - q) if np.isfinite(c) else Fraction(10) ** 400,
                                           int(np.array(c, F32).view(np.uint32)) & 1))
    return F32(best)


def _smoothstep(e0, e1, x):
    t = (x - e0) / (e1 - e0)
    t = min(max(t, F32(0.0)), F32(1.0))
    return t * t * (F32(3.0) - F32(2.0) * t)


def _fract(x):
    return x - F32(np.floor(x))


_GLSL = {
    4: _lift1(lambda a: F32(abs(a))),
    8: _lift1(lambda a: F32(np.floor(a))),
    10: _lift1(_fract),
    13: _lift1(lambda a: F32(np.sin(a))),
    14: _lift1(lambda a: F32(np.cos(a))),
    25: _lift2(lambda y, x: F32(np.arctan2(y, x))),
    26: _lift2(lambda x, y: F32(np.power(x, y))),
    27: _lift1(lambda a: F32(np.exp(a))),
    28: _lift1(lambda a: F32(np.log(a))),
    29: _lift1(lambda a: F32(np.exp2(a))),
    30: _lift1(lambda a: F32(np.log2(a))),
    31: _lift1(lambda a: F32(np.sqrt(a))),
    37: _lift2(lambda a, b: b if b < a else a),          # FMin: y < x ? y : x
    39: _lift2(lambda a, b: min(a, b)),                  # SMin
    40: _lift2(lambda a, b: b if a < b else a),          # FMax: x < y ? y : x
    42: _lift2(lambda a, b: max(a, b)),                  # SMax
    43: _lift3(lambda x, lo, hi: min(max(x, lo), hi)),   # FClamp
    45: _lift3(lambda x, lo, hi: min(max(x, lo), hi)),   # SClamp
    46: _lift3(lambda x, y, a: x * (F32(1.0) - a) + y * a),   # FMix: x*(1-a) + y*a
    49: _lift3(_smoothstep),
    50: _lift3(_fma32),
}


def _length(v):
    if not isinstance(v, list):
        return F32(abs(v))
    s = v[0] * v[0]
    for e in v[1:]:
        s = s + e * e
    return F32(np.sqrt(s))


class Module:
    def __init__(self, path: str):
        raw = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(raw) // 4), raw)
        if w[0] != 0x07230203:
            raise ValueError("not a SPIR-V module")
        self.types, self.consts, self.names, self.member_names = {}, {}, {}, {}
        self.globals, self.global_storage, self.functions, self.decor = {}, {}, {}, {}
        self.entry = None
        i, cur = 5, None
        while i < len(w):
            op, n = w[i] & 0xFFFF, w[i] >> 16
            a = w[i + 1:i + n]
            if op == 5:
                self.names[a[0]] = self._string(a[1:])
            elif op == 6:
                self.member_names[(a[0], a[1])] = self._string(a[2:])
            elif op == 15:
                self.entry = a[1]
            elif op == 71:
                self.decor.setdefault(a[0], []).append(tuple(a[1:]))
            elif op == 19:
                self.types[a[0]] = ("void",)
            elif op == 20:
                self.types[a[0]] = ("bool",)
            elif op == 21:
                self.types[a[0]] = ("int", a[1], a[2])
            elif op == 22:
                self.types[a[0]] = ("float", a[1])
            elif op == 23:
                self.types[a[0]] = ("vec", a[1], a[2])
            elif op == 25:
                self.types[a[0]] = ("image",)
            elif op == 28:
                self.types[a[0]] = ("array", a[1], a[2])
            elif op == 29:
                self.types[a[0]] = ("rtarray", a[1])
            elif op == 30:
                self.types[a[0]] = ("struct", list(a[1:]))
            elif op == 32:
                self.types[a[0]] = ("ptr", a[1], a[2])
            elif op == 33:
                self.types[a[0]] = ("func",)
            elif op == 41:
                self.consts[a[1]] = True
            elif op == 42:
                self.consts[a[1]] = False
            elif op == 43:
                t = self.types[a[0]]
                if t[0] == "float":
                    self.consts[a[1]] = F32(struct.unpack("<f", struct.pack("<I", a[2]))[0])
                else:
                    self.consts[a[1]] = _i32(a[2]) if t[2] else a[2]
            elif op == 44:
                self.consts[a[1]] = [self.consts[c] for c in a[2:]]
            elif op == 1:                                  # OpUndef
                self.consts[a[1]] = self.default(a[0])
            elif op == 59 and cur is None:                 # module-scope OpVariable
                self.globals[a[1]] = a[0]
                self.global_storage[a[1]] = a[2]
            elif op == 54:
                cur = {"id": a[1], "params": [], "blocks": {}, "order": [], "ret": a[0]}
                self.functions[a[1]] = cur
                block = None
            elif op == 55:
                cur["params"].append(a[1])
            elif op == 56:
                cur = None
            elif cur is not None:
                if op == 248:
                    block = []
                    cur["blocks"][a[0]] = block
                    cur["order"].append(a[0])
                else:
                    block.append((op, a))
            i += n

    @staticmethod
    def _string(words):
        return b"".join(struct.pack("<I", x) for x in words).split(b"\0")[0].decode()

    def default(self, tid):
        t = self.types[tid]
        if t[0] == "float":
            return F32(0.0)
        if t[0] == "int":
            return 0
        if t[0] == "bool":
            return False
        if t[0] == "vec":
            return [self.default(t[1]) for _ in range(t[2])]
        if t[0] == "struct":
            return [self.default(m) for m in t[1]]
        if t[0] == "array":
            return [self.default(t[1]) for _ in range(self.consts[t[2]])]
        if t[0] == "rtarray":
            return []
        return None

    def global_named(self, name):
        for gid in self.globals:
            if self.names.get(gid) == name:
                return gid
        raise KeyError(name)


class Invocation:
    """Runs the entry point once.  `globals_` maps module-scope variable ids to Cells the caller filled
    (push constants, built-ins, buffers); image stores are appended to `self.stores`; `probe` lists
    "function:variable" debug names (OpName) whose last stored value is kept in `self.probes`."""

    def __init__(self, module: Module, globals_: dict, image_size=(1, 1), probe=()):
        self.m = module
        self.g = globals_
        self.image_size = image_size
        self.stores = []
        self.probe = set(probe)
        self.probes = {}
        self.steps = 0

    def run(self):
        with np.errstate(all="ignore"):
            self.call(self.m.entry, [])
        return self

    # -- helpers ---------------------------------------------------------------------------------
    def val(self, env, i):
        if i in env:
            return env[i]
        if i in self.m.consts:
            return self.m.consts[i]
        if i in self.g:
            return Ptr(self.g[i])
        raise KeyError("id %%%d" % i)

    def call(self, fid, args):
        f = self.m.functions[fid]
        env = dict(zip(f["params"], args))
        fname = self.m.names.get(fid, "").split("(")[0]
        label, prev = f["order"][0], None
        blocks = f["blocks"]
        m, val = self.m, self.val
        while True:
            nxt = None
            for op, a in blocks[label]:
                self.steps += 1
                if op == 61:                                       # Load
                    env[a[1]] = val(env, a[2]).load()
                elif op == 62:                                     # Store
                    p, x = val(env, a[0]), val(env, a[1])
                    p.store(x)
                    nm = m.names.get(a[0])
                    if nm is not None and self.probe:
                        key = fname + ":" + nm
                        if key in self.probe:
                            self.probes[key] = _copy(x)
                elif op == 65:                                     # AccessChain
                    base = val(env, a[2])
                    idx = tuple(int(val(env, k)) for k in a[3:])
                    env[a[1]] = Ptr(base.cell, base.path + idx)
                elif op == 59:                                     # function-local Variable
                    t = m.types[a[0]]
                    c = Cell(m.default(t[2]))
                    if len(a) > 3:
                        c.value = _copy(val(env, a[3]))
                    env[a[1]] = Ptr(c)
                elif op == 133:
                    env[a[1]] = _MUL(val(env, a[2]), val(env, a[3]))
                elif op == 129:
                    env[a[1]] = _ADD(val(env, a[2]), val(env, a[3]))
                elif op == 131:
                    env[a[1]] = _SUB(val(env, a[2]), val(env, a[3]))
                elif op == 136:
                    env[a[1]] = _DIV(val(env, a[2]), val(env, a[3]))
                elif op == 127:
                    env[a[1]] = _lift1(lambda x: F32(-x))(val(env, a[2]))
                elif op == 142:                                    # VectorTimesScalar
                    s = val(env, a[3])
                    env[a[1]] = [e * s for e in val(env, a[2])]
                elif op == 148:                                    # Dot
                    x, y = val(env, a[2]), val(env, a[3])
                    s = x[0] * y[0]
                    for k in range(1, len(x)):
                        s = s + x[k] * y[k]
                    env[a[1]] = s
                elif op == 12:                                     # ExtInst (GLSL.std.450)
                    inst = a[3]
                    ops = [val(env, k) for k in a[4:]]
                    if inst == 66:
                        env[a[1]] = _length(ops[0])
                    elif inst == 67:
                        env[a[1]] = _length(_SUB(ops[0], ops[1]))
                    elif inst in _GLSL:
                        env[a[1]] = _GLSL[inst](*ops)
                    else:
                        raise NotImplementedError("GLSL.std.450 instruction %d" % inst)
                elif op == 80:                                     # CompositeConstruct
                    out = []
                    for k in a[2:]:
                        v = val(env, k)
                        out.extend(v) if isinstance(v, list) and m.types[a[0]][0] == "vec" else out.append(v)
                    env[a[1]] = out
                elif op == 81:                                     # CompositeExtract
                    v = val(env, a[2])
                    for k in a[3:]:
                        v = v[k]
                    env[a[1]] = _copy(v)
                elif op == 79:                                     # VectorShuffle
                    both = list(val(env, a[2])) + list(val(env, a[3]))
                    env[a[1]] = [both[k] for k in a[4:]]
                elif op == 83:
                    env[a[1]] = _copy(val(env, a[2]))
                elif op == 110:                                    # ConvertFToS (truncation)
                    env[a[1]] = _lift1(lambda x: int(np.trunc(x)) if np.isfinite(x) else 0)(val(env, a[2]))
                elif op in (111, 112):                             # ConvertSToF / ConvertUToF
                    env[a[1]] = _lift1(lambda x: F32(x))(val(env, a[2]))
                elif op == 124:                                    # Bitcast (uvec <-> ivec here)
                    t = m.types[a[0]]
                    comp = m.types[t[1]] if t[0] == "vec" else t
                    signed = comp[0] == "int" and comp[2] == 1
                    env[a[1]] = _lift1(lambda x: _i32(int(x)) if signed else int(x) & 0xFFFFFFFF)(val(env, a[2]))
                elif op == 128:
                    env[a[1]] = _lift2(lambda x, y: _i32(x + y))(val(env, a[2]), val(env, a[3]))
                elif op == 130:
                    env[a[1]] = _lift2(lambda x, y: _i32(x - y))(val(env, a[2]), val(env, a[3]))
                elif op == 132:
                    env[a[1]] = _lift2(lambda x, y: _i32(x * y))(val(env, a[2]), val(env, a[3]))
                elif op == 126:
                    env[a[1]] = _lift1(lambda x: _i32(-x))(val(env, a[2]))
                elif op == 135:                                    # SDiv (truncating)
                    env[a[1]] = _lift2(lambda x, y: _i32(int(x / y)) if y else 0)(val(env, a[2]), val(env, a[3]))
                elif op in _CMP:
                    env[a[1]] = _lift2(_CMP[op])(val(env, a[2]), val(env, a[3]))
                elif op == 168:
                    env[a[1]] = _lift1(lambda x: not x)(val(env, a[2]))
                elif op == 169:                                    # Select
                    env[a[1]] = _lift3(lambda c, x, y: x if c else y)(val(env, a[2]), val(env, a[3]), val(env, a[4]))
                elif op == 154:
                    env[a[1]] = any(val(env, a[2]))
                elif op == 155:
                    env[a[1]] = all(val(env, a[2]))
                elif op == 245:                                    # Phi
                    for k in range(2, len(a), 2):
                        if a[k + 1] == prev:
                            env[a[1]] = _copy(val(env, a[k]))
                            break
                    else:
                        raise RuntimeError("phi without a matching predecessor")
                elif op == 57:                                     # FunctionCall
                    env[a[1]] = self.call(a[2], [val(env, k) for k in a[3:]])
                elif op == 104:                                    # ImageQuerySize
                    env[a[1]] = [int(self.image_size[0]), int(self.image_size[1])]
                elif op == 99:                                     # ImageWrite
                    self.stores.append((a[0], _copy(val(env, a[1])), _copy(val(env, a[2]))))
                elif op in (246, 247, 8, 317):                     # LoopMerge / SelectionMerge / Line / NoLine
                    pass
                elif op == 249:
                    nxt = a[0]
                elif op == 250:
                    nxt = a[1] if val(env, a[0]) else a[2]
                elif op == 251:                                    # Switch
                    sel = int(val(env, a[0]))
                    nxt = a[1]
                    for k in range(2, len(a), 2):
                        if _i32(a[k]) == sel:
                            nxt = a[k + 1]
                            break
                elif op == 253:
                    return None
                elif op == 254:
                    return _copy(val(env, a[0]))
                elif op in (252, 255):
                    raise RuntimeError("reached OpKill / OpUnreachable")
                else:
                    raise NotImplementedError("opcode %d" % op)
            if nxt is None:
                raise RuntimeError("block without terminator")
            prev, label = label, nxt


_MUL = _lift2(lambda x, y: x * y)
_ADD = _lift2(lambda x, y: x + y)
_SUB = _lift2(lambda x, y: x - y)
_DIV = _lift2(lambda x, y: x / y)
_CMP = {
    164: lambda x, y: x == y, 165: lambda x, y: x != y, 166: lambda x, y: x or y, 167: lambda x, y: x and y,
    170: lambda x, y: x == y, 171: lambda x, y: x != y,
    172: lambda x, y: (x & 0xFFFFFFFF) > (y & 0xFFFFFFFF), 174: lambda x, y: (x & 0xFFFFFFFF) >= (y & 0xFFFFFFFF),
    176: lambda x, y: (x & 0xFFFFFFFF) < (y & 0xFFFFFFFF), 178: lambda x, y: (x & 0xFFFFFFFF) <= (y & 0xFFFFFFFF),
    173: lambda x, y: x > y, 175: lambda x, y: x >= y, 177: lambda x, y: x < y, 179: lambda x, y: x <= y,
    180: lambda x, y: bool(x == y), 182: lambda x, y: bool(x != y), 184: lambda x, y: bool(x < y),
    186: lambda x, y: bool(x > y), 188: lambda x, y: bool(x <= y), 190: lambda x, y: bool(x >= y),
    181: lambda x, y: not (x != y), 183: lambda x, y: not (x == y), 185: lambda x, y: not (x >= y),
    187: lambda x, y: not (x <= y), 189: lambda x, y: not (x > y), 191: lambda x, y: not (x < y),
}

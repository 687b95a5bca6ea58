"""spirv_interp.py — a small SPIR-V interpreter for exactly the opcodes the reference's precompiled shaders use
(/root/reference/shaders/shader.{rgen,rchit,rmiss}.spv, shader_shadow.rmiss.spv).  TEST INFRASTRUCTURE ONLY.

Why: the reference ships no test, no golden image and cannot run here, but it DOES ship the compiled form of its shading
code.  Executing those binaries — read as data, one invocation at a time — pins the oracle's restatement of
src/shader.rgen:61-186 and src/shader.rchit:50-96 (control flow, operand order, constants) to a reference-held artefact.
What the driver does for the reference is bound from outside: OpTraceRayKHR (acceleration-structure traversal),
OpImageSampleExplicitLod (the cube sampler) and OpImageWrite (the storage image) call back into the harness
(tests/golden/make_spirv_fixtures.py binds them to the oracle's trace / sample_sky and a frame buffer).

Arithmetic: OpFAdd/FSub/FMul/FDiv/FNegate, OpDot and the conversions are IEEE binary32, one rounding per operation, left
to right, no contraction (the literal reading of the module: glslang emitted no NoContraction decoration, so a driver MAY
fuse).  The matrix products and the GLSL.std.450 instructions (Sin, Pow, Normalize, Length, Reflect, Sqrt, Fract, FMin,
FMax) are evaluated in binary64 from their binary32 operands and rounded once — the "ideal" implementation the GLSL
precision rules allow every driver to differ from by a few ulp.  Nothing here calls the oracle's arithmetic.

Only what the four modules need is implemented; an unknown opcode raises.
"""
import math
import struct

import numpy as np

F32 = np.float32


def f32(x):
    return F32(x)


# ---- opcode numbers (SPIR-V 1.6 unified spec) ------------------------------------------------------------------------------
OP = dict(
    Source=3, SourceExtension=4, Name=5, MemberName=6, Extension=10, ExtInstImport=11, ExtInst=12, MemoryModel=14,
    EntryPoint=15, ExecutionMode=16, Capability=17, TypeVoid=19, TypeBool=20, TypeInt=21, TypeFloat=22, TypeVector=23, TypeMatrix=24,
    TypeImage=25, TypeSampler=26, TypeSampledImage=27, TypeArray=28, TypeRuntimeArray=29, TypeStruct=30, TypePointer=32,
    TypeFunction=33, ConstantTrue=41, ConstantFalse=42, Constant=43, ConstantComposite=44, Function=54, FunctionParameter=55,
    FunctionEnd=56, FunctionCall=57, Variable=59, Load=61, Store=62, AccessChain=65, Decorate=71, MemberDecorate=72,
    VectorShuffle=79, CompositeConstruct=80, CompositeExtract=81, ImageSampleExplicitLod=88, ImageWrite=99, ConvertSToF=111,
    ConvertUToF=112, Bitcast=124, FNegate=127, IAdd=128, FAdd=129, FSub=131, IMul=132, FMul=133, FDiv=136, VectorTimesScalar=142,
    VectorTimesMatrix=144, MatrixTimesVector=145, Dot=148, LogicalNot=168, Select=169, IEqual=170, ULessThan=176,
    ULessThanEqual=178, FOrdLessThan=184, FOrdGreaterThan=186, FOrdGreaterThanEqual=190, LoopMerge=246, SelectionMerge=247,
    Label=248, Branch=249, BranchConditional=250, Return=253, ReturnValue=254, TraceRayKHR=4445, TypeAccelerationStructureKHR=5341)
NAME = {v: k for k, v in OP.items()}
# GLSL.std.450
GLSL = {10: "Fract", 13: "Sin", 26: "Pow", 31: "Sqrt", 37: "FMin", 40: "FMax", 66: "Length", 69: "Normalize", 71: "Reflect"}
# BuiltIn decorations used by the ray-tracing stages
BUILTIN = {5319: "LaunchIdKHR", 5320: "LaunchSizeKHR", 5327: "InstanceCustomIndexKHR", 7: "PrimitiveId", 5330: "ObjectToWorldKHR",
           5331: "WorldToObjectKHR", 6: "InstanceId", 5321: "WorldRayOriginKHR", 5322: "WorldRayDirectionKHR", 5332: "HitTNV"}
STORAGE = {0: "UniformConstant", 2: "Uniform", 7: "Function", 12: "StorageBuffer", 5338: "RayPayloadKHR", 5339: "HitAttributeKHR",
           5342: "IncomingRayPayloadKHR", 1: "Input", 6: "Private", 4: "Workgroup"}


def _string(words):
    b = b"".join(struct.pack("<I", w) for w in words)
    return b.split(b"\0", 1)[0].decode()


class Cell:
    """storage behind an OpVariable"""
    __slots__ = ("value",)

    def __init__(self, value=None):
        self.value = value


class Pointer:
    __slots__ = ("cell", "path")

    def __init__(self, cell, path=()):
        self.cell, self.path = cell, path

    def load(self):
        v = self.cell.value
        for i in self.path:
            v = v[i]
        return v

    def store(self, x):
        if not self.path:
            self.cell.value = x
            return
        v = self.cell.value
        for i in self.path[:-1]:
            v = v[i]
        v[self.path[-1]] = x


class Module:
    def __init__(self, data):
        words = struct.unpack("<%dI" % (len(data) // 4), data)
        if words[0] != 0x07230203:
            raise ValueError("not a SPIR-V module")
        self.version, self.generator, self.bound = words[1], words[2], words[3]
        self.insts = []          # (opcode, operands tuple)
        i = 5
        while i < len(words):
            op, n = words[i] & 0xFFFF, words[i] >> 16
            if n == 0:
                raise ValueError("zero-length instruction")
            self.insts.append((op, words[i + 1:i + n]))
            i += n
        self.names, self.types, self.consts = {}, {}, {}
        self.decor, self.member_decor = {}, {}
        self.glsl_set = None
        self.entry = None        # (execution model, function id, name, interface ids)
        self.globals = {}        # id -> (pointer type id, storage class)
        self.functions = {}      # id -> dict(params=[ids], first=index of first instruction after OpFunction, labels={id: index})
        self._scan()

    # -- static pass ----------------------------------------------------------------------------------------------------------
    def _scan(self):
        cur = None
        for idx, (op, w) in enumerate(self.insts):
            if op == OP["Name"]:
                self.names[w[0]] = _string(w[1:])
            elif op == OP["ExtInstImport"]:
                if _string(w[1:]) == "GLSL.std.450":
                    self.glsl_set = w[0]
            elif op == OP["EntryPoint"]:
                name_words = w[2:]
                nm = _string(name_words)
                used = len(nm) // 4 + 1
                self.entry = (w[0], w[1], nm, tuple(w[2 + used:]))
            elif op == OP["Decorate"]:
                self.decor.setdefault(w[0], {})[w[1]] = w[2:]
            elif op == OP["MemberDecorate"]:
                self.member_decor.setdefault((w[0], w[1]), {})[w[2]] = w[3:]
            elif op == OP["TypeVoid"]:
                self.types[w[0]] = ("void",)
            elif op == OP["TypeBool"]:
                self.types[w[0]] = ("bool",)
            elif op == OP["TypeInt"]:
                self.types[w[0]] = ("int", w[1], w[2])
            elif op == OP["TypeFloat"]:
                self.types[w[0]] = ("float", w[1])
            elif op == OP["TypeVector"]:
                self.types[w[0]] = ("vector", w[1], w[2])
            elif op == OP["TypeMatrix"]:
                self.types[w[0]] = ("matrix", w[1], w[2])      # column type, column count
            elif op == OP["TypeImage"]:
                self.types[w[0]] = ("image",) + tuple(w[1:])
            elif op == OP["TypeSampledImage"]:
                self.types[w[0]] = ("sampled_image", w[1])
            elif op == OP["TypeRuntimeArray"]:
                self.types[w[0]] = ("runtime_array", w[1])
            elif op == OP["TypeArray"]:
                self.types[w[0]] = ("array", w[1], w[2])
            elif op == OP["TypeStruct"]:
                self.types[w[0]] = ("struct",) + tuple(w[1:])
            elif op == OP["TypePointer"]:
                self.types[w[0]] = ("pointer", w[1], w[2])      # storage class, pointee
            elif op == OP["TypeFunction"]:
                self.types[w[0]] = ("function",) + tuple(w)
            elif op == OP["TypeAccelerationStructureKHR"]:
                self.types[w[0]] = ("accel",)
            elif op == OP["ConstantTrue"]:
                self.consts[w[1]] = True
            elif op == OP["ConstantFalse"]:
                self.consts[w[1]] = False
            elif op == OP["Constant"]:
                t = self.types[w[0]]
                if t[0] == "float":
                    self.consts[w[1]] = np.frombuffer(struct.pack("<I", w[2]), np.float32)[0]
                else:
                    v = w[2]
                    if t[2] and v & 0x80000000:
                        v -= 1 << 32
                    self.consts[w[1]] = v
            elif op == OP["ConstantComposite"]:
                self.consts[w[1]] = [self.consts[c] for c in w[2:]]
            elif op == OP["Variable"] and cur is None:
                self.globals[w[1]] = (w[0], w[2])
            elif op == OP["Function"]:
                cur = dict(params=[], first=idx + 1, labels={}, result_type=w[0])
                self.functions[w[1]] = cur
            elif op == OP["FunctionParameter"]:
                cur["params"].append(w[1])
            elif op == OP["Label"] and cur is not None:
                cur["labels"][w[0]] = idx
            elif op == OP["FunctionEnd"]:
                cur = None

    def builtin_of(self, var_id):
        d = self.decor.get(var_id, {})
        if 11 in d:                      # Decoration BuiltIn
            return BUILTIN.get(d[11][0], "builtin%d" % d[11][0])
        return None

    def binding_of(self, var_id):
        d = self.decor.get(var_id, {})
        return d[33][0] if 33 in d else None      # Decoration Binding

    def location_of(self, var_id):
        d = self.decor.get(var_id, {})
        return d[30][0] if 30 in d else None      # Decoration Location

    def zero(self, type_id):
        """default value of a type (Function-storage variables start undefined in SPIR-V; zeros keep runs deterministic)"""
        t = self.types[type_id]
        if t[0] == "float":
            return F32(0)
        if t[0] == "int":
            return 0
        if t[0] == "bool":
            return False
        if t[0] == "vector":
            return [self.zero(t[1]) for _ in range(t[2])]
        if t[0] == "matrix":
            return [self.zero(t[1]) for _ in range(t[2])]
        if t[0] == "struct":
            return [self.zero(m) for m in t[1:]]
        if t[0] == "array":
            return [self.zero(t[1]) for _ in range(self.consts[t[2]])]
        return None

    # -- a readable listing (debugging aid) -------------------------------------------------------------------------------------
    def disassemble(self):
        out = []
        for op, w in self.insts:
            n = NAME.get(op, "Op%d" % op)
            if op in (OP["Name"], OP["MemberName"], OP["Source"], OP["SourceExtension"], OP["Extension"], OP["Capability"], OP["MemoryModel"]):
                continue
            ws = []
            for x in w:
                if x in self.names and x < self.bound:
                    ws.append("%%%d(%s)" % (x, self.names[x]))
                elif x in self.consts and not isinstance(self.consts[x], list):
                    ws.append("%%%d=%r" % (x, self.consts[x]))
                else:
                    ws.append(str(x))
            if op == OP["ExtInst"]:
                ws[3] = GLSL.get(w[3], str(w[3]))
            out.append("%-22s %s" % (n, " ".join(ws)))
        return "\n".join(out)


def _is_seq(v):
    return isinstance(v, (list, tuple))


def _map1(f, a):
    return [f(x) for x in a] if _is_seq(a) else f(a)


def _map2(f, a, b):
    if _is_seq(a):
        return [f(x, y) for x, y in zip(a, b)]
    return f(a, b)


def _u32(x):
    return int(x) & 0xFFFFFFFF


def _s32(x):
    x = int(x) & 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


class Invocation:
    """One shader invocation.  `env` supplies everything outside the module:
       env.resource(module, var_id, storage, binding, builtin, location) -> initial value of a global (or None)
       env.trace_ray(inv, flags, cull_mask, sbt_offset, sbt_stride, miss_index, origin, tmin, direction, tmax, payload_ptr)
       env.sample(inv, image, coord, lod) -> [r, g, b, a]
       env.image_write(inv, image, coord, texel)
    Globals are created once per invocation; `shared` lets a caller alias a global to an existing Cell (the incoming
    payload of a closest-hit / miss invocation is the caller's payload variable)."""

    def __init__(self, module, env, shared=None):
        self.m, self.env = module, env
        self.globals = {}
        self.steps = 0
        self.watch = None      # {variable id: name}: env.on_store(name, value) is called after every OpStore to it
        shared = shared or {}
        for vid, (ptype, storage) in module.globals.items():
            if vid in shared:
                self.globals[vid] = Pointer(shared[vid])
                continue
            pointee = module.types[ptype][2]
            init = env.resource(module, vid, STORAGE.get(storage, storage), module.binding_of(vid), module.builtin_of(vid), module.location_of(vid))
            self.globals[vid] = Pointer(Cell(init if init is not None else module.zero(pointee)))

    def run(self):
        self.call(self.m.entry[1], [])

    # ------------------------------------------------------------------------------------------------------------------------
    def call(self, fid, args):
        m = self.m
        fn = m.functions[fid]
        vals = dict(zip(fn["params"], args))
        insts = m.insts

        def V(i):
            if i in vals:
                return vals[i]
            if i in m.consts:
                return m.consts[i]
            if i in self.globals:
                return self.globals[i]
            raise KeyError("id %%%d has no value" % i)

        pc = fn["first"]
        while True:
            op, w = insts[pc]
            pc += 1
            self.steps += 1
            if op == OP["Label"] or op == OP["LoopMerge"] or op == OP["SelectionMerge"] or op == OP["FunctionParameter"]:
                continue
            if op == OP["Branch"]:
                pc = fn["labels"][w[0]]
                continue
            if op == OP["BranchConditional"]:
                pc = fn["labels"][w[1] if V(w[0]) else w[2]]
                continue
            if op == OP["Return"]:
                return None
            if op == OP["ReturnValue"]:
                return V(w[0])
            if op == OP["FunctionEnd"]:
                raise RuntimeError("fell off the end of a function")
            if op == OP["Variable"]:
                pointee = m.types[w[0]][2]
                vals[w[1]] = Pointer(Cell(V(w[3]) if len(w) > 3 else m.zero(pointee)))
            elif op == OP["Load"]:
                v = V(w[2]).load()
                vals[w[1]] = list(v) if isinstance(v, list) else v       # a loaded composite is a value, not an alias
                if isinstance(v, list) and v and isinstance(v[0], list):
                    vals[w[1]] = [list(c) for c in v]
            elif op == OP["Store"]:
                v = V(w[1])
                if isinstance(v, list):
                    v = [list(c) if isinstance(c, list) else c for c in v]
                V(w[0]).store(v)
                if self.watch and w[0] in self.watch:      # stores to named variables the harness wants to see
                    self.env.on_store(self.watch[w[0]], v)
            elif op == OP["AccessChain"]:
                base = V(w[2])
                vals[w[1]] = Pointer(base.cell, base.path + tuple(int(V(i)) for i in w[3:]))
            elif op == OP["FunctionCall"]:
                vals[w[1]] = self.call(w[2], [V(a) for a in w[3:]])
            elif op == OP["CompositeConstruct"]:
                out = []
                t = m.types[w[0]]
                for c in w[2:]:
                    v = V(c)
                    if t[0] == "vector" and _is_seq(v):
                        out.extend(v)
                    else:
                        out.append(v)
                vals[w[1]] = out
            elif op == OP["CompositeExtract"]:
                v = V(w[2])
                for i in w[3:]:
                    v = v[i]
                vals[w[1]] = v
            elif op == OP["VectorShuffle"]:
                both = list(V(w[2])) + list(V(w[3]))
                vals[w[1]] = [both[i] for i in w[4:]]
            elif op == OP["ConvertSToF"]:
                vals[w[1]] = _map1(lambda x: F32(_s32(x)), V(w[2]))
            elif op == OP["ConvertUToF"]:
                vals[w[1]] = _map1(lambda x: F32(_u32(x)), V(w[2]))
            elif op == OP["Bitcast"]:
                t = m.types[w[0]]
                st = t if t[0] != "vector" else m.types[t[1]]
                if st[0] == "int":
                    conv = (lambda x: _s32(x)) if st[2] else (lambda x: _u32(x))
                    src = V(w[2])
                    if isinstance(src, (np.floating, float)) or (_is_seq(src) and isinstance(src[0], (np.floating, float))):
                        conv = (lambda x: int(np.float32(x).view(np.uint32))) if not st[2] else (lambda x: int(np.float32(x).view(np.int32)))
                    vals[w[1]] = _map1(conv, src)
                else:
                    vals[w[1]] = _map1(lambda x: np.uint32(_u32(x)).view(np.float32), V(w[2]))
            elif op == OP["FNegate"]:
                vals[w[1]] = _map1(lambda x: F32(-x), V(w[2]))
            elif op == OP["IAdd"]:
                signed = self._signed(w[0])
                vals[w[1]] = _map2(lambda a, b: (_s32 if signed else _u32)(int(a) + int(b)), V(w[2]), V(w[3]))
            elif op == OP["IMul"]:
                signed = self._signed(w[0])
                vals[w[1]] = _map2(lambda a, b: (_s32 if signed else _u32)(int(a) * int(b)), V(w[2]), V(w[3]))
            elif op == OP["FAdd"]:
                vals[w[1]] = _map2(lambda a, b: F32(a) + F32(b), V(w[2]), V(w[3]))
            elif op == OP["FSub"]:
                vals[w[1]] = _map2(lambda a, b: F32(a) - F32(b), V(w[2]), V(w[3]))
            elif op == OP["FMul"]:
                vals[w[1]] = _map2(lambda a, b: F32(a) * F32(b), V(w[2]), V(w[3]))
            elif op == OP["FDiv"]:
                vals[w[1]] = _map2(lambda a, b: F32(a) / F32(b), V(w[2]), V(w[3]))
            elif op == OP["VectorTimesScalar"]:
                s = F32(V(w[3]))
                vals[w[1]] = [F32(x) * s for x in V(w[2])]
            elif op == OP["Dot"]:
                # literal binary32: products rounded, summed left to right, no contraction.  (The jitter hash of
                # src/shader.rgen:57-59 is chaotic in the last bit of this dot product — its value feeds sin() at ~1e5 rad —
                # so a fused or wider dot gives unrelated sub-pixel positions; SURVEY.md §8c trap 2.)
                a, b = V(w[2]), V(w[3])
                acc = F32(a[0]) * F32(b[0])
                for x, y in zip(a[1:], b[1:]):
                    acc = acc + F32(x) * F32(y)
                vals[w[1]] = acc
            elif op == OP["MatrixTimesVector"]:
                M, v = V(w[2]), V(w[3])                     # M = list of columns
                rows = len(M[0])
                vals[w[1]] = [F32(sum(float(M[c][r]) * float(v[c]) for c in range(len(M)))) for r in range(rows)]
            elif op == OP["VectorTimesMatrix"]:
                v, M = V(w[2]), V(w[3])
                vals[w[1]] = [F32(sum(float(v[r]) * float(M[c][r]) for r in range(len(v)))) for c in range(len(M))]
            elif op == OP["LogicalNot"]:
                vals[w[1]] = _map1(lambda x: not x, V(w[2]))
            elif op == OP["Select"]:
                c, a, b = V(w[2]), V(w[3]), V(w[4])
                vals[w[1]] = [x if k else y for k, x, y in zip(c, a, b)] if _is_seq(c) else (a if c else b)
            elif op == OP["IEqual"]:
                vals[w[1]] = _map2(lambda a, b: _u32(a) == _u32(b), V(w[2]), V(w[3]))
            elif op == OP["ULessThan"]:
                vals[w[1]] = _map2(lambda a, b: _u32(a) < _u32(b), V(w[2]), V(w[3]))
            elif op == OP["ULessThanEqual"]:
                vals[w[1]] = _map2(lambda a, b: _u32(a) <= _u32(b), V(w[2]), V(w[3]))
            elif op == OP["FOrdLessThan"]:
                vals[w[1]] = _map2(lambda a, b: bool(F32(a) < F32(b)), V(w[2]), V(w[3]))
            elif op == OP["FOrdGreaterThan"]:
                vals[w[1]] = _map2(lambda a, b: bool(F32(a) > F32(b)), V(w[2]), V(w[3]))
            elif op == OP["FOrdGreaterThanEqual"]:
                vals[w[1]] = _map2(lambda a, b: bool(F32(a) >= F32(b)), V(w[2]), V(w[3]))
            elif op == OP["ExtInst"]:
                if w[2] != m.glsl_set:
                    raise NotImplementedError("extended instruction set %d" % w[2])
                vals[w[1]] = self._glsl(GLSL.get(w[3], w[3]), [V(a) for a in w[4:]])
            elif op == OP["TraceRayKHR"]:
                a = [V(x) for x in w]
                self.env.trace_ray(self, _u32(a[1]), _u32(a[2]), _u32(a[3]), _u32(a[4]), _u32(a[5]), a[6], a[7], a[8], a[9], a[10])
            elif op == OP["ImageSampleExplicitLod"]:
                # operands: result type, result, sampled image, coordinate, image-operands mask (Lod = 0x2), lod
                lod = V(w[5]) if len(w) > 5 and (w[4] & 0x2) else F32(0)
                vals[w[1]] = self.env.sample(self, V(w[2]), V(w[3]), lod)
            elif op == OP["ImageWrite"]:
                self.env.image_write(self, V(w[0]), V(w[1]), V(w[2]))
            else:
                raise NotImplementedError("opcode %s" % NAME.get(op, op))

    def _signed(self, type_id):
        t = self.m.types[type_id]
        if t[0] == "vector":
            t = self.m.types[t[1]]
        return bool(t[2])

    @staticmethod
    def _glsl(name, a):
        if name == "Fract":
            return _map1(lambda x: F32(F32(x) - F32(math.floor(float(x)))), a[0])
        if name == "Sin":
            return _map1(lambda x: F32(math.sin(float(x))), a[0])
        if name == "Sqrt":
            return _map1(lambda x: F32(np.sqrt(F32(x))), a[0])
        if name == "Pow":
            return _map2(lambda x, y: F32(math.pow(float(x), float(y))) if not (float(x) == 0.0 and float(y) <= 0.0) else F32(float("nan")), a[0], a[1])
        if name == "FMin":
            return _map2(lambda x, y: y if F32(y) < F32(x) else x, a[0], a[1])
        if name == "FMax":
            return _map2(lambda x, y: y if F32(x) < F32(y) else x, a[0], a[1])
        if name == "Length":
            v = a[0] if _is_seq(a[0]) else [a[0]]
            return F32(math.sqrt(sum(float(x) * float(x) for x in v)))
        if name == "Normalize":
            v = a[0] if _is_seq(a[0]) else [a[0]]
            l = math.sqrt(sum(float(x) * float(x) for x in v))
            out = [F32(float(x) / l) for x in v]
            return out if _is_seq(a[0]) else out[0]
        if name == "Reflect":
            I, N = a
            d = sum(float(n) * float(i) for n, i in zip(N, I))
            return [F32(float(i) - 2.0 * d * float(n)) for i, n in zip(I, N)]
        raise NotImplementedError("GLSL.std.450 instruction %s" % name)

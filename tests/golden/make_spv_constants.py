#!/usr/bin/env python3
"""Extract the scalar constants of the reference's precompiled SPIR-V shaders.

The reference ships `shaders/shader.{rgen,rchit,rmiss}.spv` and
`shaders/shader_shadow.rmiss.spv` (glslang output of `src/shader.*`).  They are
the only machine-checked artefact of the shading path the reference holds, so
the bit patterns of their OpConstant words are used as golden vectors for the
oracle's constants (jitter hash coefficients, tmin/tmax, 2.5 focal term,
ambient = Iamb*ka folded by glslang, ior and 1/ior, 0.9, 100, 0.01 ...).

This script only READS the binaries as data (SPIR-V word stream, opcode 43 =
OpConstant, 22 = OpTypeFloat, 21 = OpTypeInt); nothing is executed.

Run in the authoring container (needs /root/reference):
    python tests/golden/make_spv_constants.py
writes tests/golden/spv_constants.json
"""
import json
import os
import struct
import sys

REF = os.environ.get("RT_REFERENCE", "/root/reference")
FILES = ["shader.rgen.spv", "shader.rchit.spv", "shader.rmiss.spv", "shader_shadow.rmiss.spv"]


def constants(path):
    data = open(path, "rb").read()
    words = struct.unpack("<%dI" % (len(data) // 4), data)
    assert words[0] == 0x07230203, "not SPIR-V"
    types = {}
    f32, i32 = [], []
    i = 5
    while i < len(words):
        op, n = words[i] & 0xFFFF, words[i] >> 16
        if op == 22:  # OpTypeFloat
            types[words[i + 1]] = ("f", words[i + 2])
        elif op == 21:  # OpTypeInt
            types[words[i + 1]] = ("i", words[i + 2], words[i + 3])
        elif op == 43:  # OpConstant
            t = types.get(words[i + 1])
            if t and t[0] == "f" and t[1] == 32:
                f32.append("0x%08x" % words[i + 3])
            elif t and t[0] == "i" and t[1] == 32:
                i32.append(words[i + 3])
        i += n
    return {"version": "0x%08x" % words[1], "generator": "0x%08x" % words[2], "f32_bits": f32, "u32": i32}


def main():
    out = {f: constants(os.path.join(REF, "shaders", f)) for f in FILES}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spv_constants.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    sys.exit(main())

"""Extracts the 32-bit float constants of the reference's COMPILED shaders (shaders/*.comp.spv, SPIR-V
binaries shipped in the reference tree) into tests/golden/spv_constants.json.

This is the one artefact of the reference's GPU path that can be read without a Vulkan stack: every
literal the GLSL compiler kept (palette knots and break points, tonemap and luma coefficients, the folded
log(2.0) and 1.0/2.2) is an OpConstant of a float type.  tests/test_oracle_kat.py checks that every literal
typed into the oracle's and the library's palette / post-chain code is one of them.
Run (needs /root/reference):  python tests/golden/make_spv_constants.py
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
SHADERS = "/root/reference/FractalRenderer/shaders"


def float_constants(path):
    raw = open(path, "rb").read()
    w = struct.unpack("<%dI" % (len(raw) // 4), raw)
    assert w[0] == 0x07230203, "not SPIR-V"
    float_types, out, i = {}, set(), 5
    while i < len(w):
        op, n = w[i] & 0xFFFF, w[i] >> 16
        if op == 22:                                   # OpTypeFloat  %id  width
            float_types[w[i + 1]] = w[i + 2]
        elif op == 43 and float_types.get(w[i + 1]) == 32:   # OpConstant  %type  %id  value
            out.add(w[i + 3])
        i += n
    return sorted(out)


def main():
    data = {}
    for name in ("mandelbrot", "julia", "burning_ship", "test_deep_zoom"):
        bits = float_constants(os.path.join(SHADERS, name + ".comp.spv"))
        data[name] = {"float32_bits": bits,
                      "values": [struct.unpack("<f", struct.pack("<I", b))[0] for b in bits]}
    json.dump(data, open(os.path.join(HERE, "spv_constants.json"), "w"), indent=1)
    print({k: len(v["float32_bits"]) for k, v in data.items()})


if __name__ == "__main__":
    main()

"""Minimal PNG reader for the tests (colour type RGB, filter None, no interlace): verifies the
signature and every chunk CRC, returns (pixels, chunks)."""
import struct
import zlib

import numpy as np


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks, idat = 8, [], b""
    while pos < len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        crc, = struct.unpack(">I", data[pos + 8 + ln:pos + 12 + ln])
        assert crc == (zlib.crc32(typ + body) & 0xFFFFFFFF), typ
        chunks.append((typ.decode(), body))
        if typ == b"IDAT":
            idat += body
        pos += 12 + ln
    assert chunks[0][0] == "IHDR" and chunks[-1][0] == "IEND"
    w, h, depth, ctype, comp, flt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (ctype, comp, flt, inter) == (2, 0, 0, 0) and depth in (8, 16)
    raw = zlib.decompress(idat)
    row = w * 3 * depth // 8
    assert len(raw) == (row + 1) * h
    rows = np.frombuffer(raw, np.uint8).reshape(h, row + 1)
    assert np.all(rows[:, 0] == 0)                       # filter type None
    px = rows[:, 1:]
    if depth == 8:
        return px.reshape(h, w, 3).copy(), chunks
    return px.reshape(h, w, 3, 2).astype(np.uint16).dot(np.array([256, 1], np.uint16)).astype(np.uint16), chunks

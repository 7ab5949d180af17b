"""Reader/writer for the tiny tagged-array container ("WAF1") the checkers exchange.

Layout: b"WAF1" then records  [u32 name_len][name][dtype char][u64 count][raw little-endian data]
dtype chars: i=int32 q=int64 f=float32 d=float64 B=uint8.
Written by oracle/ref_harness.cpp (the real reference) and by tests/golden/make_golden.py.
"""
import struct

import numpy as np

_DT = {"i": np.int32, "q": np.int64, "f": np.float32, "d": np.float64, "B": np.uint8}
_CH = {np.dtype(v): k for k, v in _DT.items()}


def load(path):
    data = open(path, "rb").read()
    assert data[:4] == b"WAF1", path
    off, out = 4, {}
    while off < len(data):
        (nl,) = struct.unpack_from("<I", data, off)
        off += 4
        name = data[off:off + nl].decode()
        off += nl
        ch = chr(data[off])
        off += 1
        (cnt,) = struct.unpack_from("<Q", data, off)
        off += 8
        dt = np.dtype(_DT[ch])
        out[name] = np.frombuffer(data, dtype=dt, count=cnt, offset=off).copy()
        off += cnt * dt.itemsize
    return out


def save(path, arrays):
    with open(path, "wb") as f:
        f.write(b"WAF1")
        for name, arr in arrays.items():
            arr = np.ascontiguousarray(arr)
            ch = _CH[arr.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(ch.encode())
            f.write(struct.pack("<Q", arr.size))
            f.write(arr.tobytes())


def scalar(d, k):
    return d[k].reshape(-1)[0].item()


def text(d, k):
    return d[k].tobytes().decode("latin-1")

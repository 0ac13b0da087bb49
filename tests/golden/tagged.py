"""Tagged binary array container shared by oracle/ref_driver.cpp and the golden
fixture generator.  Record = name_len u32 | name | dtype char | ndim u32 |
dims u64[ndim] | raw little-endian data.  dtype: 'f' float32, 'l' int64, 'B' uint8.
"""
import struct

import numpy as np

_DT = {"f": np.float32, "l": np.int64, "B": np.uint8}
_RT = {np.dtype(np.float32): "f", np.dtype(np.int64): "l", np.dtype(np.uint8): "B"}


def write_tagged(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            t = _RT[a.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + t.encode())
            f.write(struct.pack("<I", a.ndim))
            for s in a.shape:
                f.write(struct.pack("<Q", s))
            f.write(a.tobytes())


def read_tagged(path):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    p = 0
    while p < len(buf):
        (nl,) = struct.unpack_from("<I", buf, p)
        p += 4
        name = buf[p:p + nl].decode()
        p += nl
        t = chr(buf[p])
        p += 1
        (nd,) = struct.unpack_from("<I", buf, p)
        p += 4
        dims = struct.unpack_from("<%dQ" % nd, buf, p)
        p += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        dt = np.dtype(_DT[t])
        out[name] = np.frombuffer(buf, dtype=dt, count=n, offset=p).reshape(dims).copy()
        p += n * dt.itemsize
    return out

"""Reader for the '.slud' record container written by oracle/ref/slu_ref_dump.c."""
import struct
import numpy as np

_DT = {0: np.int32, 1: np.int64, 2: np.float64, 3: np.complex128}


def read_slud(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    p = 0
    while p < len(data):
        (nl,) = struct.unpack_from("<i", data, p); p += 4
        name = data[p:p + nl].decode(); p += nl
        dt, cnt = struct.unpack_from("<iq", data, p); p += 12
        dtype = np.dtype(_DT[dt])
        arr = np.frombuffer(data, dtype=dtype, count=cnt, offset=p).copy(); p += cnt * dtype.itemsize
        out[name] = arr
    return out

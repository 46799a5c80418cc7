"""Reader / writer of the RRXB container used by the C++ drivers in place of NetCDF (include_test/Netcdf_interface.h has
the format). A variable is (array, [dimension names]); arrays are stored in C order, exactly the NetCDF data model."""
import struct
import numpy as np

_DT = {0: np.float64, 1: np.float32, 2: np.int32, 3: np.int8}
_ID = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int32): 2, np.dtype(np.int8): 3}


def write(path, dims, variables):
    with open(path, "wb") as f:
        f.write(b"RRXB1\0\0\0")
        f.write(struct.pack("<I", len(dims)))
        for n, s in dims.items():
            b = n.encode(); f.write(struct.pack("<I", len(b))); f.write(b); f.write(struct.pack("<q", int(s)))
        f.write(struct.pack("<I", len(variables)))
        for n, (arr, dnames) in variables.items():
            arr = np.asarray(arr)
            arr = np.ascontiguousarray(arr).reshape(arr.shape)
            if arr.dtype == np.int64:
                arr = arr.astype(np.int32)
            if arr.dtype.kind == "S":
                arr = arr.view(np.int8)
            assert arr.dtype in _ID, (n, arr.dtype)
            exp = tuple(dims[d] for d in dnames)
            assert arr.shape == exp, (n, arr.shape, exp)
            b = n.encode(); f.write(struct.pack("<I", len(b))); f.write(b)
            f.write(struct.pack("<B", _ID[arr.dtype])); f.write(struct.pack("<I", len(dnames)))
            for d in dnames:
                db = d.encode(); f.write(struct.pack("<I", len(db))); f.write(db)
            f.write(struct.pack("<q", arr.nbytes)); f.write(arr.tobytes())


def read(path):
    with open(path, "rb") as f:
        assert f.read(8)[:5] == b"RRXB1"
        def u32(): return struct.unpack("<I", f.read(4))[0]
        def s(): return f.read(u32()).decode()
        dims = {}
        for _ in range(u32()):
            n = s(); dims[n] = struct.unpack("<q", f.read(8))[0]
        variables = {}
        for _ in range(u32()):
            n = s(); dt = struct.unpack("<B", f.read(1))[0]
            dn = [s() for _ in range(u32())]
            nb = struct.unpack("<q", f.read(8))[0]
            arr = np.frombuffer(f.read(nb), dtype=_DT[dt]).reshape([dims[d] for d in dn]).copy()
            variables[n] = (arr, dn)
    return dims, variables


def strings(names, length=32):
    out = np.zeros((len(names), length), dtype=np.int8)
    for i, n in enumerate(names):
        b = n.encode().ljust(length, b" ")
        out[i] = np.frombuffer(b, dtype=np.int8)
    return out

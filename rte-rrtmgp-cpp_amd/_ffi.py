"""Minimal ctypes marshalling shared by the HIP binding (this package) and the test-only oracle bindings.

A ``Lib`` wraps one shared object and one calling convention:
  by_ref=False : scalars by value (librrx_hip.so, include/rrx_hip.h; also oracle/_ref)
  by_ref=True  : every scalar passed by address (the Fortran bind(C) convention of include/rrtmgp_kernels.h)
Arguments: int -> c_int, float -> Float, BoolArg -> signed char, numpy array / torch tensor -> data pointer,
None -> NULL, ctypes objects are passed through.
"""
import ctypes
import numpy as np


class BoolArg:
    """Marks a scalar that must be marshalled as the reference's Bool (signed char)."""
    def __init__(self, value):
        self.value = 1 if value else 0


class PtrArg:
    """A raw pointer value (e.g. a hipStream_t)."""
    def __init__(self, value):
        self.value = int(value) if value else 0


class Lib:
    def __init__(self, path, float_dtype, by_ref=False, returns_status=False, error_fn=None):
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.float_dtype = np.dtype(float_dtype)
        self.cfloat = ctypes.c_double if self.float_dtype == np.float64 else ctypes.c_float
        self.by_ref = by_ref
        self.returns_status = returns_status
        self.error_fn = error_fn

    def has(self, name):
        return hasattr(self.cdll, name)

    def _marshal(self, a, keep):
        if a is None:
            return ctypes.c_void_p(0)
        if isinstance(a, BoolArg):
            v = ctypes.c_byte(a.value)
        elif isinstance(a, PtrArg):
            return ctypes.c_void_p(a.value)
        elif isinstance(a, (bool, np.bool_)):
            raise TypeError("wrap booleans in BoolArg")
        elif isinstance(a, (int, np.integer)):
            v = ctypes.c_int(int(a))
        elif isinstance(a, (float, np.floating)):
            v = self.cfloat(float(a))
        elif isinstance(a, np.ndarray):
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("array arguments must be contiguous")
            keep.append(a)
            return ctypes.c_void_p(a.ctypes.data)
        elif hasattr(a, "data_ptr"):                     # torch tensor
            if not a.is_contiguous():
                raise ValueError("tensor arguments must be contiguous")
            keep.append(a)
            return ctypes.c_void_p(a.data_ptr())
        elif isinstance(a, ctypes._SimpleCData) or isinstance(a, ctypes.Array):
            return a
        else:
            raise TypeError(f"cannot marshal {type(a)}")
        if self.by_ref:
            keep.append(v)
            return ctypes.byref(v)
        return v

    def call(self, name, *args):
        fn = getattr(self.cdll, name)
        fn.restype = ctypes.c_int if self.returns_status else None
        keep = []
        cargs = [self._marshal(a, keep) for a in args]
        rc = fn(*cargs)
        if self.returns_status and rc != 0:
            msg = self.error_fn() if self.error_fn else ""
            raise RuntimeError(f"{name} failed (status {rc}): {msg}")
        return rc

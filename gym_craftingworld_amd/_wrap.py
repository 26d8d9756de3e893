"""Zero-copy torch views of engine-owned device buffers (cw_buffers pointers).

torch is plumbing here (device tensors, streams): the engine owns the memory for its lifetime,
the tensors are non-owning views -- the reference's live-alias contract (ray.py:194-196).
The DLPack capsule (device type kDLROCM) is produced by the C library (cwh_dlpack_make: struct and
deleter live in C, so dropping the last view never calls back into Python -- it may happen during
interpreter shutdown); if torch refuses it, the __cuda_array_interface__ route is tried.
"""
import ctypes as C

import torch

from . import _lib as L

_KDL_INT, _KDL_UINT = 0, 1
_DTYPES = {
    torch.uint8: (_KDL_UINT, 8, '|u1'), torch.int8: (_KDL_INT, 8, '|i1'), torch.int16: (_KDL_INT, 16, '<i2'),
    torch.int32: (_KDL_INT, 32, '<i4'), torch.int64: (_KDL_INT, 64, '<i8'),
}

C.pythonapi.PyCapsule_New.restype = C.py_object
C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]


def _via_dlpack(ptr, shape, dtype, device_index):
    code, bits, _ = _DTYPES[dtype]
    shp = (C.c_int64 * len(shape))(*shape)
    m = L.load().cwh_dlpack_make(C.c_void_p(ptr), device_index, code, bits, len(shape), shp)
    if not m:
        raise MemoryError('cwh_dlpack_make')
    cap = C.pythonapi.PyCapsule_New(m, b'dltensor', None)
    return torch.from_dlpack(cap)      # the consumer now owns the managed tensor and calls its C deleter


class _CAI:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def tensor_view(ptr, shape, dtype, device_index):
    """Non-owning torch tensor over device memory at `ptr`."""
    if not ptr:
        return None
    shape = tuple(int(s) for s in shape)
    try:
        t = _via_dlpack(int(ptr), shape, dtype, device_index)
    except Exception:  # noqa: BLE001
        t = torch.as_tensor(_CAI(ptr, shape, _DTYPES[dtype][2]), device='cuda:%d' % device_index)
    if t.data_ptr() != int(ptr) or tuple(t.shape) != shape:
        raise RuntimeError('zero-copy wrap failed (got a copy)')
    return t

"""Zero-copy torch views of engine-owned device buffers (cw_buffers pointers).

torch is plumbing here (device tensors, streams): the engine owns the memory for its lifetime,
the tensors are non-owning views -- the reference's live-alias contract (ray.py:194-196).
A DLPack capsule is built with ctypes (device type kDLROCM); if torch refuses it, the
__cuda_array_interface__ route is tried.
"""
import ctypes as C

import torch

_KDL_ROCM = 10
_KDL_INT, _KDL_UINT, _KDL_BOOL = 0, 1, 6


class _DLDevice(C.Structure):
    _fields_ = [('device_type', C.c_int32), ('device_id', C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [('code', C.c_uint8), ('bits', C.c_uint8), ('lanes', C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [('data', C.c_void_p), ('device', _DLDevice), ('ndim', C.c_int32), ('dtype', _DLDataType),
                ('shape', C.POINTER(C.c_int64)), ('strides', C.POINTER(C.c_int64)), ('byte_offset', C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [('dl_tensor', _DLTensor), ('manager_ctx', C.c_void_p), ('deleter', _DELETER)]

_live = {}  # address of DLManagedTensor -> (struct, shape array) kept alive until torch calls the deleter


@_DELETER
def _deleter(ptr):
    _live.pop(C.addressof(ptr.contents), None)


_DTYPES = {
    torch.uint8: (_KDL_UINT, 8, '|u1'), torch.int8: (_KDL_INT, 8, '|i1'), torch.int16: (_KDL_INT, 16, '<i2'),
    torch.int32: (_KDL_INT, 32, '<i4'), torch.int64: (_KDL_INT, 64, '<i8'),
}

C.pythonapi.PyCapsule_New.restype = C.py_object
C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]


def _via_dlpack(ptr, shape, dtype, device_index):
    code, bits, _ = _DTYPES[dtype]
    m = _DLManagedTensor()
    shp = (C.c_int64 * len(shape))(*shape)
    m.dl_tensor.data = ptr
    m.dl_tensor.device = _DLDevice(_KDL_ROCM, device_index)
    m.dl_tensor.ndim = len(shape)
    m.dl_tensor.dtype = _DLDataType(code, bits, 1)
    m.dl_tensor.shape = shp
    m.dl_tensor.strides = None
    m.dl_tensor.byte_offset = 0
    m.manager_ctx = None
    m.deleter = _deleter
    _live[C.addressof(m)] = (m, shp)
    cap = C.pythonapi.PyCapsule_New(C.addressof(m), b'dltensor', None)
    try:
        return torch.from_dlpack(cap)
    except Exception:
        _live.pop(C.addressof(m), None)
        raise


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
    except Exception:
        t = torch.as_tensor(_CAI(ptr, shape, _DTYPES[dtype][2]), device='cuda:%d' % device_index)
    if t.data_ptr() != int(ptr) or tuple(t.shape) != shape:
        raise RuntimeError('zero-copy wrap failed (got a copy)')
    return t

"""warp.types as far as warp_utils.py:246-330 (torch2warp_*) needs it -- TEST INFRASTRUCTURE ONLY."""
import ctypes

import numpy as np

float32 = np.float32


def array(ptr=None, dtype=None, shape=None, copy=False, owner=False, requires_grad=False, device=None, **_):
    """`warp.types.array(ptr=tensor.data_ptr(), dtype=..., shape=n)`: a Warp array over a float32 torch buffer.  The values
    are read through the pointer; the interpreter keeps them in its arithmetic type, so the array does not alias the
    tensor (the reference's callers clone before importing and never read the tensor again)."""
    import warp as wp
    per = {wp.vec3: 3, wp.mat33: 9, wp.quat: 4}.get(dtype, 1)
    n = int(shape) if not isinstance(shape, (tuple, list)) else int(np.prod(shape))
    raw = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n * per,)).copy()
    elem = {wp.vec3: (3,), wp.mat33: (3, 3), wp.quat: (4,)}.get(dtype, ())
    kind = dtype if dtype in (wp.vec3, wp.mat33, wp.quat) else float
    return wp.array(raw.astype(wp.DT).reshape((n,) + elem), kind)

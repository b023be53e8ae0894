"""isx -- Python binding of libisx.so (hand-written HIP kernels for MI355X / gfx950).

`isx.ops` exposes one function per C-ABI entry of include/isx.h, taking torch CUDA
tensors.  There is NO CPU fallback: every op raises if the HIP library is missing or
a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401

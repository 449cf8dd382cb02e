"""kyber_amd -- host-side Python binding of libkyber_hip.so, the MI355X (gfx950)
batched round-3 Kyber KEM engine.

The product is the C-ABI library (include/kyber_batch.h, include/kyber_api.h);
this package is only a ctypes mirror of it for tests, smoke() and bench.py.
There is no CPU fallback: importing works anywhere, but every compute call
goes to the HIP library and fails loudly if it is missing or no GPU is present.
"""
from .lib import (  # noqa: F401
    KyberHip,
    KyberHipError,
    MlKemHip,
    PARAMETER_SETS,
    library,
    library_path,
    sizes,
)

__all__ = ["KyberHip", "KyberHipError", "MlKemHip", "PARAMETER_SETS", "library", "library_path", "sizes"]

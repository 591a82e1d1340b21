"""libultrahdr_dev_amd -- MI355X (gfx950) native Ultra HDR gain-map pixel path.

The product is the C-ABI shared library ``libuhdr_hip.so`` (include/uhdr_hip.h) built from the
hand-written HIP kernels in ``csrc/``.  This Python package is only the ctypes binding used by the
tests and the benchmark; importing :mod:`libultrahdr_dev_amd.api` fails loudly when the library
has not been built (``python -m libultrahdr_dev_amd.build``) -- there is no CPU fallback.
"""
__all__ = ["api", "build"]

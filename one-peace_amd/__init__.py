"""ONE-PEACE hot path, MI355X-native.

Host-side mirrors of the reference's operator/model interface (same class names, parameter names,
forward signatures) over a C-ABI HIP library (``lib/libonepeace_hip.so``, declared in
``include/onepeace_hip.h``).  See DESIGN.md for the path and its boundary.
"""
__version__ = "0.1.0"

from . import hip  # noqa: F401  (ctypes binding; loading is lazy)

"""MI355X-native path-tracing core: Python plumbing over the C ABI of libptcore.so.

The product is the shared library built from ``pathtracer_amd/csrc`` (HIP kernels for gfx950
and the C++ host code); this package only loads it with ctypes for tests, the benchmark and
the multi-GPU driver.  Importing the package never falls back to a CPU renderer: if the
library is missing, :func:`pathtracer_amd.lib.load` raises.
"""
from .lib import load, PtError, Params, Stats  # noqa: F401
from .renderer import Renderer  # noqa: F401

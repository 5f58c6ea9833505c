"""tools/ only: `import _toollib` before anything touches lama_amd's library makes LAMA_TOOL_LIB=<path> (default: none = the product build) the
process-wide library through lama_amd._lib.use_library -- e.g. the profiling build, whose kernel-selection switches read LAMA_* variables.
The product binding itself reads no environment variable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_p = os.environ.get('LAMA_TOOL_LIB')
if _p:
    from lama_amd import _lib as _L
    _L.use_library(os.path.abspath(_p))

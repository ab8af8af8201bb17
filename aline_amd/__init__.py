"""aline_amd -- MI355X-native (gfx950) implementation of ALINE's amortized inference-and-design
inner loop behind the reference's module interface.  Importing the package loads
`csrc/libaline_hip.so`; there is no CPU fallback."""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .model import Aline, Embedder, Encoder, OutputHead  # noqa: F401

__version__ = "0.1.0"

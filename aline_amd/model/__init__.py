from .base import Aline
from .embedder import Embedder
from .encoder import Encoder
from .head import OutputHead, AcquisitionHead, GMMTargetHead

__all__ = ["Aline", "Embedder", "Encoder", "OutputHead", "AcquisitionHead", "GMMTargetHead"]

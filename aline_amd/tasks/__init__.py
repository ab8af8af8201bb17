from .base_task import Task
from .location_finding import HiddenLocation

__all__ = ["Task", "HiddenLocation"]

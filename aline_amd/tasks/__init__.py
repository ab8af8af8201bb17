from .base_task import Task
from .location_finding import HiddenLocation
from .ces import CESTask

__all__ = ["Task", "HiddenLocation", "CESTask"]

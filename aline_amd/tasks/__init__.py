from .base_task import Task
from .location_finding import HiddenLocation
from .ces import CESTask
from .gaussian_process import GPTask
from .psychometric import PsychometricTask

__all__ = ["Task", "HiddenLocation", "CESTask", "GPTask", "PsychometricTask"]

from .base_ds import BaseDS
from .file_ds import RoboMove, RoboMoveSimple, SpringNonlinear
from .file_ds import Sarcos, Actuator, Ballbeam, Dryer, Drive, Furnace
from .synthetic_ds import make_synthetic_ds

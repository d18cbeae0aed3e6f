"""Datasets of the reference's run scripts (`from cbfssm.datasets import Sarcos, Actuator, ...`) plus a synthetic one of
any shape for tests and benchmarks."""
from . import base_ds as _base, file_ds as _files, synthetic_ds as _synthetic

BaseDS = _base.BaseDS
make_synthetic_ds = _synthetic.make_synthetic_ds
_NAMES = ('Actuator', 'Ballbeam', 'Drive', 'Dryer', 'Furnace', 'RoboMove', 'RoboMoveSimple', 'Sarcos', 'SpringNonlinear')
globals().update({name: getattr(_files, name) for name in _NAMES})
__all__ = ['BaseDS', 'make_synthetic_ds'] + list(_NAMES)

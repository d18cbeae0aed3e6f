"""cbfssm.model: the CBF-SSM model class on the MI355X HIP path.

CBFSSM, its forward-only variant CBFSSMHALF and the PR-SSM baseline run on the HIP kernels.  The reference also ships
Voliro (cbfssm/model/__init__.py:4: a hexacopter physics model on a proprietary dataset); it is outside the hot path
this build accelerates and raises a clear error instead of silently falling back."""
from .cbfssm import CBFSSM
from .cbfssmhalf import CBFSSMHALF
from .prssm import PRSSM
from .session import Session, OutOfRangeError, InvalidArgumentError


def _not_built(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is not part of the MI355X hot-path build (only cbfssm.model.CBFSSM is); '
                                      'see DESIGN.md, "out of scope"' % name)
    _Missing.__name__ = name
    return _Missing


Voliro = _not_built('Voliro')

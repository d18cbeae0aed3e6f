"""cbfssm.model: the CBF-SSM model class on the MI355X HIP path.

CBFSSM and its forward-only variant CBFSSMHALF run on the HIP kernels.  The reference also ships PRSSM and Voliro
(cbfssm/model/__init__.py:3-4); they are outside the hot path this build accelerates and raise
a clear error instead of silently falling back."""
from .cbfssm import CBFSSM
from .cbfssmhalf import CBFSSMHALF
from .session import Session, OutOfRangeError, InvalidArgumentError


def _not_built(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is not part of the MI355X hot-path build (only cbfssm.model.CBFSSM is); '
                                      'see DESIGN.md, "out of scope"' % name)
    _Missing.__name__ = name
    return _Missing


PRSSM = _not_built('PRSSM')
Voliro = _not_built('Voliro')

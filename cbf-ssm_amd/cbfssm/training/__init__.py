"""Training loop with the reference's call surface (`from cbfssm.training import Trainer`)."""
from . import trainer as _trainer

Trainer = _trainer.Trainer
__all__ = ['Trainer']

"""Minimal stand-ins for the TensorFlow objects the reference's callers hold (tf.Graph, tf.Session, tf.train.Saver,
tf.errors.OutOfRangeError) so that training/trainer.py- and outputs/outputs.py-shaped code keeps its call shapes:

    with model.graph.as_default():
        with Session() as sess:
            sess.run(model.init); model.load_ds(sess, ...); model.run(sess, (model.train, model.loss), {...})

A Fetch is a symbolic handle; Session.run hands the fetch list to the model that owns the handles, which executes
ONE mini-batch on the HIP path -- exactly one `sess.run` of the reference (model/base_model.py:42-69).
"""
import contextlib
import os
import torch


class OutOfRangeError(Exception):
    """End of the mini-batch iterator (tf.errors.OutOfRangeError, base_model.py:64)."""


class InvalidArgumentError(RuntimeError):
    """Cholesky of a non positive definite K_mm (what TensorFlow raises from tf.cholesky)."""


class Fetch:
    def __init__(self, model, name):
        self.model, self.name = model, name

    def __repr__(self):
        return '<Fetch %s>' % self.name


class Graph:
    @contextlib.contextmanager
    def as_default(self):
        yield self


class Session:
    """Holds the device; one process drives one GPU (LOCAL_RANK picks it under torch.distributed.run)."""

    def __init__(self, device=None, config=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError('cbfssm needs an MI355X: no HIP device is visible and there is no CPU fallback')
            device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', '0'))
        self.device = torch.device(device)
        self._init_process_group()

    def _init_process_group(self):
        """Under `python -m torch.distributed.run ... run/run_*.py` (WORLD_SIZE > 1 in the environment) the first
        Session brings up the process group: one process per GPU, backend nccl = RCCL over xGMI (CBFSSM_DIST_BACKEND
        overrides: the tests use gloo)."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world <= 1 or self.device.type != 'cuda':
            return
        import torch.distributed as td
        if not td.is_available() or td.is_initialized():
            return
        torch.cuda.set_device(self.device)
        backend = os.environ.get('CBFSSM_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            td.init_process_group('nccl', device_id=self.device)
        else:
            td.init_process_group(backend)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (tuple, list))
        flist = [fetches] if single else list(fetches)
        models = {f.model for f in flist if isinstance(f, Fetch)}
        assert len(models) == 1, 'fetches must belong to one model'
        model = models.pop()
        feed = {}
        for k, v in (feed_dict or {}).items():
            feed[k.name if isinstance(k, Fetch) else k] = v
        res = model._execute(self, [f.name for f in flist], feed)
        return res[0] if single else tuple(res)


class Saver:
    """best.ckpt / model.ckpt semantics of tf.train.Saver (training/trainer.py:30-31,58-63): the 12 trainable
    tensors plus the Adam slots, written with torch.save to the same path stems."""

    def __init__(self, model):
        self.model = model

    def save(self, sess, path):
        # data parallel: the ranks hold identical parameters; rank 0 writes, the others wait until the file is there
        from ..hip.dist_utils import is_writer, barrier
        if is_writer():
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(self.model._state_dict(), path)
        barrier()

    def restore(self, sess, path):
        sd = torch.load(path, map_location='cpu', weights_only=True)
        self.model._load_state_dict(sess, sd)

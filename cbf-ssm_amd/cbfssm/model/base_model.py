"""Host-side data pipeline and run loop of a model: the counterpart of cbfssm/model/base_model.py:6-69.

tf.data semantics kept: from_tensor_slices -> repeat(repeats) -> shuffle(buffer) -> batch(batch_size, keep the
partial last batch) -> one mini-batch per `sess.run` until OutOfRange (base_model.py:24-31,42-69).
"""
import os
import sys
import numpy as np

from .session import Fetch, Graph, OutOfRangeError


def shuffle_order(n, buffer_size, rng):
    """Element order produced by tf.data's shuffle(buffer_size) over n elements: a sliding reservoir."""
    if buffer_size <= 1:
        return np.arange(n)
    order, buf, nxt = [], [], 0
    while nxt < n and len(buf) < buffer_size:
        buf.append(nxt)
        nxt += 1
    while buf:
        i = int(rng.integers(len(buf)))
        order.append(buf[i])
        if nxt < n:
            buf[i] = nxt
            nxt += 1
        else:
            buf[i] = buf[-1]
            buf.pop()
    return np.asarray(order, dtype=np.int64)


class BaseModel:

    def __init__(self, config, dtype='float64'):
        self.config = config
        self.dtype = dtype
        self.graph = Graph()
        self.condition = Fetch(self, 'condition')     # fed through feed_dict (base_model.py:19)
        self._rng = np.random.default_rng(config.get('seed', None))
        self._data = None
        self._order = None
        self._cursor = 0
        self._staged = None
        self._build_graph()

    def _build_graph(self):
        pass

    def load_ds(self, sess, data_in, data_out, repeats=1):
        """base_model.py:36-40: (re)initialise the iterator over [n_seq, T, dim] arrays."""
        data_in = np.asarray(data_in, dtype=np.float64)
        data_out = np.asarray(data_out, dtype=np.float64)
        assert data_in.ndim == 3 and data_out.ndim == 3 and data_in.shape[:2] == data_out.shape[:2]
        n = data_in.shape[0]
        base = np.tile(np.arange(n), int(repeats))
        self._order = base[shuffle_order(base.size, int(self.config['shuffle']), self._rng)]
        self._data = (data_in, data_out)
        self._cursor = 0
        self._staged = None                     # (a mini-batch a model staged ahead belongs to the old iterator)

    def _next_batch(self):
        if self._data is None or self._cursor >= self._order.size:
            raise OutOfRangeError()
        bs = int(self.config['batch_size'])
        idx = self._order[self._cursor:self._cursor + bs]           # partial final batch kept (base_model.py:26)
        self._cursor += bs
        return self._data[0][idx], self._data[1][idx]

    @staticmethod
    def run(sess, tensors, feed_dict, show_progress=False):
        """base_model.py:42-69: sess.run until the iterator is exhausted, results concatenated on axis 0.

        When every fetch is a scalar of the ELBO (train, loss, loglik, ...: what Trainer runs, training/trainer.py:40,46) the
        mini-batches of the pass are issued back to back and their scalars are read from the device ONCE, after the last
        one: the reference's `sess.run` hands the loss to the host every step, which on the GPU is a stream synchronisation
        per step with the device idle while the host prepares the next launch.  Same values, same order; a failed Cholesky
        (InvalidArgumentError) surfaces at the end of the pass instead of at its step."""
        flist = list(tensors) if isinstance(tensors, (tuple, list)) else [tensors]
        models = {f.model for f in flist if isinstance(f, Fetch)}
        model = models.pop() if len(models) == 1 else None
        names = [f.name for f in flist if isinstance(f, Fetch)]
        if (model is not None and len(names) == len(flist) and hasattr(model, '_SCALAR_FETCHES') and not show_progress
                and all(n in model._SCALAR_FETCHES for n in names) and not os.environ.get('CBFSSM_RUN_SYNC')):
            import torch
            feed = {(k.name if isinstance(k, Fetch) else k): v for k, v in (feed_dict or {}).items()}
            rows = []
            while True:
                try:
                    rows.append(model._execute(sess, names, feed, lazy=True))
                except OutOfRangeError:
                    break
            if not rows:
                return None
            host = torch.stack(rows).cpu().numpy()                   # the pass's one device-to-host transfer
            per_run = [model._scalar_results(names, h) for h in host]
            return [np.asarray([r[i] for r in per_run]) if n != 'train' else None for i, n in enumerate(names)]
        res_all = None
        while True:
            try:
                res = sess.run(tensors, feed_dict=feed_dict)
                if show_progress:
                    sys.stdout.write('.')
                    sys.stdout.flush()
                if not isinstance(res, tuple):
                    res = (res,)
                if res_all is None:
                    res_all = [np.atleast_1d(r) if r is not None else None for r in res]
                else:
                    for i, item in enumerate(res):
                        if item is not None:
                            res_all[i] = np.concatenate((res_all[i], np.atleast_1d(item)), axis=0)
            except OutOfRangeError:
                break
        if show_progress:
            print()
        return res_all

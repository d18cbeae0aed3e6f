"""Dataset container behind `cbfssm.datasets.*`: raw train/test arrays [experiments, time, dim], z-normalisation
statistics taken from the training split, and the windowed mini-batch arrays [windows, seq_len, dim] a model's
`load_ds` consumes.  Field and method names are the ones the reference's run scripts and `Outputs` use
(cbfssm/datasets/base_ds.py:5-85); the implementation is this package's own."""
import os
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

_SPLITS = ('train_in', 'train_out', 'test_in', 'test_out')


def _default_data_dir():
    here = os.path.dirname(os.path.abspath(__file__))
    return os.environ.get('CBFSSM_DATA_DIR', os.path.join(here, 'data'))


class BaseDS:
    dim_u = None        # set by the concrete datasets
    dim_y = None

    def __init__(self, seq_len, seq_stride):
        self.seq_len, self.seq_stride = seq_len, seq_stride
        empty = np.empty(0)
        for split in _SPLITS:
            setattr(self, split, empty)
            setattr(self, split + '_batch', empty)
        self.mean = dict.fromkeys(('in', 'out'), np.empty(()))
        self.std = dict.fromkeys(('in', 'out'), np.empty(()))
        self.data_path = _default_data_dir() + '/'

    # ---- z-normalisation (statistics of the flattened training data)
    def normalize_init(self, data_in, data_out):
        for key, arr in (('in', data_in), ('out', data_out)):
            arr = np.asarray(arr)
            if arr.ndim != 2:
                raise AssertionError('normalisation statistics are taken from [samples, dim] arrays')
            mu = arr.mean(axis=0)
            self.mean[key] = mu
            self.std[key] = (arr - mu).std(axis=0)

    def normalize(self, data, key):
        return (data - self.mean[key]) / self.std[key]

    def denormalize(self, data, key, shift=True):
        scaled = data * self.std[key]
        if shift:
            scaled = scaled + self.mean[key]
        return scaled

    # ---- windowing
    @staticmethod
    def rnn_batches(x, length, stride, _unused=0):
        """[experiments, time, dim] -> [windows, length, dim]: a window every `stride` samples of every experiment, plus
        one over the last `length` samples when the strided windows do not reach the end."""
        x = np.asarray(x)
        assert x.ndim == 3, "data must be shaped as [experiments x time x dimension]"
        n_time = x.shape[1]
        assert n_time >= length, "Sequence length must be shorter than data."
        starts = list(range(0, n_time - length + 1, stride))
        if starts[-1] + length < n_time:
            starts.append(n_time - length)
        # (experiments, time - length + 1, dim, length) view -> pick the starts -> (experiments, windows, length, dim)
        view = sliding_window_view(x, length, axis=1)[:, starts]
        return np.ascontiguousarray(np.moveaxis(view, -1, 2)).reshape(-1, length, x.shape[2])

    def get_batches(self, seq_len, seq_stride):
        return tuple(self.rnn_batches(getattr(self, split), seq_len, seq_stride) for split in _SPLITS)

    def create_batches(self):
        for split, windows in zip(_SPLITS, self.get_batches(self.seq_len, self.seq_stride)):
            setattr(self, split + '_batch', windows)
        self.print_stats()

    def print_stats(self):
        rows = [('sequence length', self.seq_len)]
        for split in ('train', 'test'):
            raw = getattr(self, split + '_in')
            rows.append((split + ' samples', raw.shape[0] * raw.shape[1]))
            rows.append((split + ' sequences', getattr(self, split + '_in_batch').shape[0]))
        print('Dataset Stats:')
        for label, value in rows:
            print('  %s: %d' % (label, value))

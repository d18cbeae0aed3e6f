"""Dataset container with the reference's fields and windowing (cbfssm/datasets/base_ds.py:5-85): z-normalisation
from the training data, sliding windows of seq_len every seq_stride with the tail window appended."""
import os
import numpy as np


class BaseDS:

    dim_u = None
    dim_y = None

    def __init__(self, seq_len, seq_stride):
        self.seq_len = seq_len
        self.seq_stride = seq_stride
        for name in ('train_in', 'train_out', 'test_in', 'test_out', 'train_in_batch', 'train_out_batch',
                     'test_in_batch', 'test_out_batch'):
            setattr(self, name, np.empty(0))
        self.mean = {'in': np.empty(()), 'out': np.empty(())}
        self.std = {'in': np.empty(()), 'out': np.empty(())}
        self.data_path = os.environ.get('CBFSSM_DATA_DIR', os.path.join(os.path.dirname(__file__), 'data')) + '/'

    def normalize_init(self, data_in, data_out):
        assert data_in.ndim == 2 and data_out.ndim == 2
        self.mean['in'] = np.mean(data_in, axis=0)
        self.std['in'] = np.std(data_in - self.mean['in'], axis=0)
        self.mean['out'] = np.mean(data_out, axis=0)
        self.std['out'] = np.std(data_out - self.mean['out'], axis=0)

    def normalize(self, data, key):
        return (data - self.mean[key]) / self.std[key]

    def denormalize(self, data, key, shift=True):
        res = data * self.std[key]
        return res + self.mean[key] if shift else res

    def get_batches(self, seq_len, seq_stride):
        return tuple(self.rnn_batches(a, seq_len, seq_stride, 0)
                     for a in (self.train_in, self.train_out, self.test_in, self.test_out))

    def create_batches(self):
        (self.train_in_batch, self.train_out_batch, self.test_in_batch,
         self.test_out_batch) = self.get_batches(self.seq_len, self.seq_stride)
        self.print_stats()

    @staticmethod
    def rnn_batches(x, length, stride, _):
        """[experiments, time, dim] -> [windows, length, dim]; the last `length` samples always form a window."""
        x = np.asarray(x)
        assert x.ndim == 3, "data must be shaped as [experiments x time x dimension]"
        out = []
        for ex in x:
            n = ex.shape[0]
            assert n >= length, "Sequence length must be shorter than data."
            out.extend(ex[i:i + length] for i in range(0, n - length + 1, stride))
            if (n - length) % stride > 0:
                out.append(ex[-length:])
        return np.stack(out, axis=0)

    def print_stats(self):
        print('Dataset Stats:')
        print('  sequence length: %d' % self.seq_len)
        print('  train samples: %d' % (self.train_in.shape[0] * self.train_in.shape[1]))
        print('  train sequences: %d' % self.train_in_batch.shape[0])
        print('  test samples: %d' % (self.test_in.shape[0] * self.test_in.shape[1]))
        print('  test sequences: %d' % self.test_in_batch.shape[0])

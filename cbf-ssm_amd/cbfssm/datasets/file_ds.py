"""The reference's dataset classes by name (cbfssm/datasets/__init__.py:1-2).  The data files are not part of the
reference repository (README.md:10-18) and there is no network here: each class reads its file from
cbfssm/datasets/data/ (or $CBFSSM_DATA_DIR) in the format the reference's loaders expect and raises FileNotFoundError
with that path otherwise.  File formats follow datasets/ds_manager.py:11-23 and datasets/prssm/real_world_tasks.py."""
import os
import numpy as np
import scipy.io

from .base_ds import BaseDS


class _FileDS(BaseDS):

    def _need(self, filename):
        path = os.path.join(self.data_path, filename)
        if not os.path.isfile(path):
            raise FileNotFoundError('%s needs %s (see the reference README for where to obtain it)'
                                    % (type(self).__name__, path))
        return path

    def _finish(self, in_train, out_train, in_test, out_test):
        """lists/arrays [experiments, time, dim] in physical units -> normalised fields + windows
        (datasets/prssm_ds.py:16-29)."""
        in_train, out_train = np.asarray(in_train, dtype=np.float64), np.asarray(out_train, dtype=np.float64)
        in_test, out_test = np.asarray(in_test, dtype=np.float64), np.asarray(out_test, dtype=np.float64)
        self.normalize_init(in_train.reshape(-1, self.dim_u), out_train.reshape(-1, self.dim_y))
        self.train_in, self.train_out = self.normalize(in_train, 'in'), self.normalize(out_train, 'out')
        self.test_in, self.test_out = self.normalize(in_test, 'in'), self.normalize(out_test, 'out')
        self.create_batches()


class _DSManagerDS(_FileDS):
    """.mat with ds_u / ds_x / ds_y written by DSManager.save_ds (datasets/dsmanager_ds.py:11-27)."""
    filename, split, y_crop = None, None, None

    def __init__(self, seq_len, seq_stride):
        super().__init__(seq_len, seq_stride)
        ds = scipy.io.loadmat(self._need(self.filename))
        u, y = ds['ds_u'].astype(np.float64), ds['ds_y'].astype(np.float64)
        if self.y_crop is not None:
            y = y[:, :self.y_crop]
        self.normalize_init(u, y)
        u, y = self.normalize(u, 'in'), self.normalize(y, 'out')
        s = self.split
        self.train_in, self.train_out = u[None, :s], y[None, :s]
        self.test_in, self.test_out = u[None, s:], y[None, s:]
        self.create_batches()


class RoboMoveSimple(_DSManagerDS):
    dim_u, dim_y, filename, split = 2, 4, 'robomove_simple.mat', 25000


class RoboMove(_DSManagerDS):
    dim_u, dim_y, filename, split = 2, 2, 'robomove.mat', 25000


class SpringNonlinear(_DSManagerDS):
    dim_u, dim_y, filename, split, y_crop = 1, 1, 'spring_nonlinear.mat', 5000, 1


class _SplitSeriesDS(_FileDS):
    """single-experiment system-identification benchmarks split at a sample index (real_world_tasks.py:95-256)."""
    dim_u, dim_y = 1, 1
    filename, split_point = None, None

    def _columns(self, path):
        raise NotImplementedError

    def __init__(self, seq_len, seq_stride):
        super().__init__(seq_len, seq_stride)
        u, y = self._columns(self._need(self.filename))
        u, y = np.asarray(u, dtype=np.float64).reshape(-1, 1), np.asarray(y, dtype=np.float64).reshape(-1, 1)
        s = self.split_point
        self._finish(u[None, :s], y[None, :s], u[None, s:], y[None, s:])


class Actuator(_SplitSeriesDS):
    filename, split_point = 'actuator.mat', 512

    def _columns(self, path):
        d = scipy.io.loadmat(path)
        return d['u'], d['p']


class Drive(_SplitSeriesDS):
    filename, split_point = 'drive.mat', 250

    def _columns(self, path):
        d = scipy.io.loadmat(path)
        return d['u1'], d['z1']


class Ballbeam(_SplitSeriesDS):
    filename, split_point = 'ballbeam.dat', 500

    def _columns(self, path):
        d = np.loadtxt(path)
        return d[:, 0], d[:, 1]


class Dryer(_SplitSeriesDS):
    filename, split_point = 'dryer.dat', 500

    def _columns(self, path):
        d = np.loadtxt(path)
        return d[:, 0], d[:, 1]


class Furnace(_SplitSeriesDS):
    filename, split_point = 'gas_furnace.csv', 148

    def _columns(self, path):
        d = np.loadtxt(path, skiprows=1, delimiter=',')
        return d[:, 0], d[:, 1]


class Sarcos(_FileDS):
    """sarcos_inv.mat: 66 experiments of 674 samples, every 2nd sample kept; inputs = the 7 torques, outputs = the
    7 joint positions; experiments 0-59 train, 60-65 test (real_world_tasks.py:30-92)."""
    dim_u, dim_y = 7, 7

    def __init__(self, seq_len, seq_stride):
        super().__init__(seq_len, seq_stride)
        data = scipy.io.loadmat(self._need('sarcos_inv.mat'))['sarcos_inv'].astype(np.float64)
        exps = [data[i:i + 674][::2] for i in range(0, data.shape[0], 674)]
        tr, te = exps[0:60], exps[60:66]
        self._finish([e[:, 21:28] for e in tr], [e[:, 0:7] for e in tr],
                     [e[:, 21:28] for e in te], [e[:, 0:7] for e in te])

"""CPU tests of the host-side mirror of the reference interface: iterator semantics, windowing, dataset classes,
checkpoint-free pieces of the model surface (nothing here needs the GPU)."""
import numpy as np
import pytest

from cbfssm.model.base_model import shuffle_order, BaseModel
from cbfssm.model.session import OutOfRangeError
from cbfssm.datasets import BaseDS, make_synthetic_ds, Sarcos, Actuator, RoboMove
from cbfssm.model import CBFSSM, CBFSSMHALF, PRSSM, Voliro
from cbfssm import synthetic as syn


def test_rnn_batches_windows_and_tail():
    x = np.arange(2 * 23 * 1, dtype=float).reshape(2, 23, 1)
    w = BaseDS.rnn_batches(x, 10, 4, 0)
    # starts 0,4,8,12 + tail window (23-10) % 4 = 1 > 0  -> 5 windows per experiment (base_ds.py:64-74)
    assert w.shape == (10, 10, 1)
    assert np.array_equal(w[4, :, 0], x[0, -10:, 0])
    w = BaseDS.rnn_batches(x[:, :22], 10, 4, 0)      # (22-10) % 4 == 0 -> no tail window
    assert w.shape == (8, 10, 1)
    with pytest.raises(AssertionError):
        BaseDS.rnn_batches(x, 30, 1, 0)


def test_shuffle_order_is_a_permutation_with_bounded_displacement():
    rng = np.random.default_rng(0)
    o = shuffle_order(100, 10, rng)
    assert sorted(o.tolist()) == list(range(100))
    # tf.data shuffle(buffer): element i cannot be emitted before position i - buffer + 1
    assert all(pos >= e - 9 for pos, e in enumerate(o))
    assert np.array_equal(shuffle_order(7, 1, rng), np.arange(7))
    o = shuffle_order(50, 10000, rng)
    assert sorted(o.tolist()) == list(range(50)) and not np.array_equal(o, np.arange(50))


def test_iterator_keeps_partial_final_batch_and_repeats():
    m = BaseModel({'batch_size': 4, 'shuffle': 1})
    a = np.arange(10 * 3 * 2, dtype=float).reshape(10, 3, 2)
    m.load_ds(None, a, a[:, :, :1])
    sizes = []
    while True:
        try:
            sizes.append(m._next_batch()[0].shape[0])
        except OutOfRangeError:
            break
    assert sizes == [4, 4, 2]
    m.load_ds(None, a, a[:, :, :1], repeats=2)
    n = 0
    while True:
        try:
            n += m._next_batch()[0].shape[0]
        except OutOfRangeError:
            break
    assert n == 20


def test_synthetic_ds_and_normalisation():
    ds = make_synthetic_ds(dim_u=2, dim_y=3, n_train=300, n_test=100)(50, 25)
    assert ds.train_in.shape == (1, 300, 2) and ds.test_out.shape == (1, 100, 3)
    assert ds.train_in_batch.shape == (11, 50, 2) and ds.test_in_batch.shape == (3, 50, 2)
    np.testing.assert_allclose(ds.train_in[0].mean(0), 0, atol=1e-12)
    np.testing.assert_allclose(ds.train_out[0].std(0), 1, atol=1e-12)
    back = ds.denormalize(ds.normalize(np.ones((4, 3)), 'out'), 'out')
    np.testing.assert_allclose(back, 1.0)


def test_file_datasets_fail_loudly_without_their_files(tmp_path, monkeypatch):
    monkeypatch.setenv('CBFSSM_DATA_DIR', str(tmp_path))
    for cls in (Sarcos, Actuator, RoboMove):
        assert cls.dim_u >= 1 and cls.dim_y >= 1
        with pytest.raises(FileNotFoundError):
            cls(50, 10)


def test_actuator_loader_reads_reference_format(tmp_path, monkeypatch):
    import scipy.io
    rng = np.random.default_rng(0)
    scipy.io.savemat(str(tmp_path / 'actuator.mat'), {'u': rng.standard_normal((1024, 1)), 'p': rng.standard_normal((1024, 1))})
    monkeypatch.setenv('CBFSSM_DATA_DIR', str(tmp_path))
    ds = Actuator(50, 1)
    assert ds.train_in.shape == (1, 512, 1) and ds.test_in.shape == (1, 512, 1)
    assert ds.train_in_batch.shape == (463, 50, 1)


def test_model_surface_without_gpu():
    w = syn.tiny()
    cfg = w.model_config()
    cfg['seed'] = 3
    m = CBFSSM(cfg)
    for name in ('graph', 'init', 'saver', 'condition', 'train', 'loss', 'pred_mean', 'pred_var', 'var_dict',
                 'load_ds', 'run'):
        assert hasattr(m, name), name
    assert set(m.var_dict) == {'process noise', 'observation noise', 'kernel lengthscales f', 'kernel variance f',
                               'IP pos f', 'IP mean f', 'IP var f', 'kernel lengthscales b', 'kernel variance b',
                               'IP pos b', 'IP mean b', 'IP var b'}
    iv = m._init_values
    assert iv['f.zeta_pos'].shape == (w.M, w.D) and iv['b.zeta_mean'].shape == (w.M, w.dim_x - w.dim_y)
    assert np.all(np.abs(iv['f.zeta_pos']) <= cfg['zeta_pos'])
    np.testing.assert_allclose(np.logaddexp(0, iv['var_x_unc']) + 1e-10, cfg['var_x'], rtol=1e-9)
    with pytest.raises(NotImplementedError):
        Voliro(cfg)
    pcfg = dict(cfg)
    pcfg['var_y'] = np.asarray([0.3 ** 2] * w.dim_y)
    pcfg['recog_model'] = 'conv'
    pcfg['recog_len'] = 16
    pm = PRSSM(pcfg)
    assert pm._init_values['recog.dense_kernel'].shape == (35, w.dim_x) and pm._init_values['lengthscales_unc'].shape == (1,)
    assert set(pm.var_dict) == {'process noise', 'observation noise', 'kernel lengthscales', 'kernel variance', 'IP pos',
                                'IP mean', 'IP var'}
    hcfg = dict(cfg)
    hcfg['var_y'] = np.asarray([0.3 ** 2] * w.dim_y)
    h = CBFSSMHALF(hcfg)
    assert h._init_values['recog.gate_kernel'].shape == (w.dim_u + w.dim_y + 16, 32)
    assert 'IP pos b' not in h.var_dict and 'IP pos f' in h.var_dict
    with m.graph.as_default():
        pass
    bad = dict(cfg)
    bad['var_x'] = np.zeros(w.dim_x)
    with pytest.raises(AssertionError):
        CBFSSM(bad)          # tf_transform.backward asserts positivity (tf_transform.py:14)


def test_every_file_loader_reads_its_reference_format(tmp_path, monkeypatch):
    """One synthetic file per dataset class in the format the reference's loaders read (datasets/prssm/
    real_world_tasks.py:99-256: keys, columns, split points, Sarcos' 674-sample experiments downsampled by two;
    datasets/ds_manager.py:11-23 + dsmanager_ds.py:11-63: ds_u / ds_x / ds_y, split 25000 / 5000, y cropped to its
    first column for SpringNonlinear): shapes and splits of the loaded arrays."""
    import scipy.io
    from cbfssm.datasets import (Actuator, Ballbeam, Drive, Dryer, Furnace, Sarcos, RoboMove, RoboMoveSimple,
                                 SpringNonlinear)
    rng = np.random.default_rng(0)
    d = tmp_path
    scipy.io.savemat(str(d / 'actuator.mat'), {'u': rng.standard_normal((1024, 1)), 'p': rng.standard_normal((1024, 1))})
    scipy.io.savemat(str(d / 'drive.mat'), {'u1': rng.standard_normal((500, 1)), 'z1': rng.standard_normal((500, 1))})
    np.savetxt(str(d / 'ballbeam.dat'), rng.standard_normal((1000, 2)))
    np.savetxt(str(d / 'dryer.dat'), rng.standard_normal((1000, 2)))
    np.savetxt(str(d / 'gas_furnace.csv'), rng.standard_normal((296, 2)), delimiter=',', header='u,y', comments='')
    sarcos = rng.standard_normal((66 * 674, 28))
    scipy.io.savemat(str(d / 'sarcos_inv.mat'), {'sarcos_inv': sarcos})
    for name, n, dy in (('robomove.mat', 30000, 2), ('robomove_simple.mat', 30000, 4), ('spring_nonlinear.mat', 6000, 2)):
        scipy.io.savemat(str(d / name), {'title': 'synthetic', 'ds_u': rng.standard_normal((n, 2 if 'robo' in name else 1)),
                                         'ds_x': rng.standard_normal((n, 3)), 'ds_y': rng.standard_normal((n, dy))})
    monkeypatch.setenv('CBFSSM_DATA_DIR', str(d))
    expect = {Actuator: (512, 512), Drive: (250, 250), Ballbeam: (500, 500), Dryer: (500, 500), Furnace: (148, 148)}
    for cls, (n_tr, n_te) in expect.items():
        ds = cls(20, 5)
        assert (cls.dim_u, cls.dim_y) == (1, 1)
        assert ds.train_in.shape == (1, n_tr, 1) and ds.train_out.shape == (1, n_tr, 1), cls.__name__
        assert ds.test_in.shape == (1, n_te, 1) and ds.test_out.shape == (1, n_te, 1), cls.__name__
        np.testing.assert_allclose(ds.train_in.reshape(-1, 1).mean(0), 0, atol=1e-12)       # z-normalised on the train split
        np.testing.assert_allclose(ds.train_out.reshape(-1, 1).std(0), 1, atol=1e-12)
    ds = Sarcos(100, 50)
    assert ds.train_in.shape == (60, 337, 7) and ds.test_in.shape == (6, 337, 7) and ds.train_out.shape == (60, 337, 7)
    # experiment 3, every second sample, torques = columns 21..27, positions = columns 0..6
    np.testing.assert_allclose(ds.denormalize(ds.train_in[3], 'in'), sarcos[3 * 674:4 * 674:2, 21:28], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ds.denormalize(ds.test_out[0], 'out'), sarcos[60 * 674:61 * 674:2, 0:7], rtol=1e-12, atol=1e-12)
    for cls, (du, dy, split, n) in {RoboMove: (2, 2, 25000, 30000), RoboMoveSimple: (2, 4, 25000, 30000),
                                    SpringNonlinear: (1, 1, 5000, 6000)}.items():
        ds = cls(50, 25)
        assert (cls.dim_u, cls.dim_y) == (du, dy)
        assert ds.train_in.shape == (1, split, du) and ds.test_out.shape == (1, n - split, dy), cls.__name__

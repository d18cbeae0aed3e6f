"""End-to-end run of the reference's run-script flow (run/template.py:50-64) against this package on the GPU:
dataset -> model -> Trainer.train -> Outputs.create_all, then a retrain (run/run_robomove.py:47,64 curriculum)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_template_flow(tmp_path):
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.training import Trainer
    from cbfssm.outputs import Outputs, OutputSummary
    from cbfssm.model import CBFSSM

    root_dir = str(tmp_path / 'exp')
    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=1)
    dim_x = 3
    model_config = {
        'ds': ds_sel, 'batch_size': 8, 'shuffle': 10000, 'seed': 5,
        'dim_x': dim_x, 'ind_pnt_num': 20, 'samples': 10, 'learning_rate': 0.05,
        'loss_factors': np.asarray([1., 0.]), 'k_factor': 5., 'recog_len': 8,
        'zeta_pos': 2., 'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2,
        'var_x': np.asarray([0.002 ** 2] * dim_x), 'var_y': np.asarray([1. ** 2] * dim_x),
        'gp_var': 0.5 ** 2, 'gp_len': 2.,
    }
    summary = OutputSummary(root_dir)
    outputs = Outputs(root_dir)
    ds = ds_sel(40, 20)
    outputs.set_ds(ds)
    model = CBFSSM(model_config)
    outputs.set_model(model, root_dir)
    trainer = Trainer(model, root_dir)
    trainer.train(ds, 4)
    outputs.set_trainer(trainer)
    outputs.create_all()
    summary.add_outputs(outputs)
    summary.write_summary()

    assert len(trainer.train_all) == 4 and all(np.isfinite(trainer.train_all)) and all(np.isfinite(trainer.test_all))
    assert trainer.train_all[-1] < trainer.train_all[0]
    for f in ('best.ckpt', 'model.ckpt', 'mse.txt', 'var_dump.txt', 'predict_train.mat', 'predict_test.mat',
              'summary.txt'):
        assert os.path.isfile(os.path.join(root_dir, f)), f
    assert np.isfinite(outputs.get_last_rmse())
    text = open(os.path.join(root_dir, 'var_dump.txt')).read()
    assert 'process noise:' in text and 'IP pos f:' in text

    # resume from model.ckpt: Adam state and parameters continue (trainer.py:30-31)
    trainer2 = Trainer(model, root_dir)
    trainer2.train(ds, 1, retrain=True)
    assert trainer2.train_all[0] < trainer.train_all[0]


def test_batched_test_experiments_equal_the_reference_loop(monkeypatch):
    """Outputs.test_mse runs all test experiments in ONE launch (CBFSSM.run_experiments) where the reference loops over
    B = 1 `sess.run`s (outputs/outputs.py:121-133).  Same model, same seed, once per mode: the per-experiment predictive
    means are bitwise equal (every experiment gets the noise the k-th run of the loop would have drawn, in the chain order
    c = b S + s) and so is the RMSE."""
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.outputs import Outputs
    from cbfssm.model import CBFSSM
    from cbfssm.model.session import Session

    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=200, n_test=400, seed=1)
    dim_x = 3
    cfg = {'ds': ds_sel, 'batch_size': 8, 'shuffle': 100, 'seed': 11, 'dim_x': dim_x, 'ind_pnt_num': 20, 'samples': 10,
           'learning_rate': 0.05, 'loss_factors': np.asarray([1., 0.]), 'k_factor': 5., 'recog_len': 8, 'zeta_pos': 2.,
           'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2, 'var_x': np.asarray([0.002 ** 2] * dim_x),
           'var_y': np.asarray([1. ** 2] * dim_x), 'gp_var': 0.5 ** 2, 'gp_len': 2.}
    ds = ds_sel(40, 20)
    # five test experiments of 80 steps (the synthetic set comes as one 400-step experiment)
    ds.test_in = np.ascontiguousarray(ds.test_in.reshape(5, 80, -1))
    ds.test_out = np.ascontiguousarray(ds.test_out.reshape(5, 80, -1))
    res = {}
    for mode in ('loop', 'batched'):
        if mode == 'loop':
            monkeypatch.setenv('CBFSSM_OUTPUTS_LOOP', '1')
        else:
            monkeypatch.delenv('CBFSSM_OUTPUTS_LOOP', raising=False)
        np.random.seed(123)                              # the unseeded numpy initialisers of _setup_vars (gp_tf.py:112-118)
        m = CBFSSM(dict(cfg))
        out = Outputs('/tmp')
        out.set_ds(ds)
        out.model = m
        preds = []
        with m.graph.as_default(), Session() as sess:
            sess.run(m.init)
            if mode == 'loop':
                for k in range(ds.test_in.shape[0]):
                    m.load_ds(sess, ds.test_in[k:k + 1], ds.test_out[k:k + 1])
                    preds.append(m.run(sess, m.pred_mean, {m.condition: False})[0])
            else:
                preds = list(m.run_experiments(sess, m.pred_mean, ds.test_in, ds.test_out, {m.condition: False}))
        # ... and through Outputs.test_mse itself (fresh model of the same seed: the noise generator starts over)
        np.random.seed(123)
        m2 = CBFSSM(dict(cfg))
        out.model = m2
        with m2.graph.as_default(), Session() as sess:
            sess.run(m2.init)
            out._path = lambda name: os.path.join('/tmp', 'cbfssm_test_' + mode + '_' + name)
            out.test_mse(sess)
        res[mode] = (preds, out.last_rmse)
    assert len(res['loop'][0]) == len(res['batched'][0])
    for a, b in zip(res['loop'][0], res['batched'][0]):
        assert np.asarray(a).shape == np.asarray(b).shape and np.array_equal(np.asarray(a), np.asarray(b))
    assert res['loop'][1] == res['batched'][1]


def test_pipelined_pass_equals_the_per_step_loop(monkeypatch):
    """`model.run` over scalar fetches issues the mini-batches of a pass back to back and reads their losses once
    (base_model.py:42-69 hands every loss to the host, a stream synchronisation per step): the same model and seed trained
    both ways gives bitwise the same per-batch losses, test losses and parameters, incl. a ragged last batch."""
    from cbfssm.datasets import make_synthetic_ds
    from cbfssm.model import CBFSSM
    from cbfssm.model.session import Session

    ds_sel = make_synthetic_ds(dim_u=1, dim_y=1, n_train=400, n_test=160, seed=1)
    dim_x = 3
    cfg = {'ds': ds_sel, 'batch_size': 7, 'shuffle': 100, 'seed': 11, 'dim_x': dim_x, 'ind_pnt_num': 20, 'samples': 10,
           'learning_rate': 0.05, 'loss_factors': np.asarray([1., 0.]), 'k_factor': 5., 'recog_len': 8, 'zeta_pos': 2.,
           'zeta_mean': 0.05 ** 2, 'zeta_var': 0.01 ** 2, 'var_x': np.asarray([0.002 ** 2] * dim_x),
           'var_y': np.asarray([1. ** 2] * dim_x), 'gp_var': 0.5 ** 2, 'gp_len': 2.}
    ds = ds_sel(40, 20)
    res = {}
    for mode in ('sync', 'pipelined'):
        if mode == 'sync':
            monkeypatch.setenv('CBFSSM_RUN_SYNC', '1')
        else:
            monkeypatch.delenv('CBFSSM_RUN_SYNC', raising=False)
        np.random.seed(321)
        m = CBFSSM(dict(cfg))
        with m.graph.as_default(), Session() as sess:
            sess.run(m.init)
            tr, te = [], []
            for _ in range(2):
                m.load_ds(sess, ds.train_in_batch, ds.train_out_batch)
                out = m.run(sess, (m.train, m.loss), {m.condition: True})
                assert out[0] is None
                tr.append(out[1])
                m.load_ds(sess, ds.test_in_batch, ds.test_out_batch)
                te.append(m.run(sess, m.loss, {m.condition: True})[0])
            pars = {k: sess.run(v) for k, v in m.var_dict.items()}
        res[mode] = (np.concatenate(tr), np.concatenate(te), pars)
    assert res['sync'][0].shape == res['pipelined'][0].shape and res['sync'][0].size > 2
    assert np.array_equal(res['sync'][0], res['pipelined'][0]) and np.array_equal(res['sync'][1], res['pipelined'][1])
    for k in res['sync'][2]:
        assert np.array_equal(res['sync'][2][k], res['pipelined'][2][k]), k

"""`Outputs(out_dir)`: evaluation artefacts of a trained model, with the method names and files of the reference's
cbfssm/outputs/outputs.py:11-164 --

    best.ckpt restored, then
    training_loss.pdf          epoch losses of the attached Trainer (needs matplotlib)
    predict_{train,test}.mat   free-running prediction (condition=False) of the first experiment, at most 300 steps:
                               denormalised mean / std / ground truth (+ .pdf with the 1.96 sigma band)
    mse.txt                    MSE and RMSE of the denormalised free-running prediction, averaged over test experiments
    var_dump.txt               every entry of model.var_dict

The prediction and RMSE run through `model.pred_mean` / `model.pred_var` with B = 1, i.e. the persistent HIP kernels.
"""
import os
import numpy as np
from scipy.io import savemat

from ..model.session import Session
from ..hip.dist_utils import is_writer

try:                                        # figures are optional: the numbers do not depend on matplotlib
    import matplotlib
    matplotlib.use('Agg')
    from matplotlib import pyplot as plt
except Exception:                           # pragma: no cover
    plt = None

_BAND_COLOUR = (1.0, 178.0 / 255.0, 110.0 / 255.0)


def _rows_as_text(value):
    """var_dump.txt layout: vectors on one line, matrices one line per row, '% .4e' per entry."""
    fmt = lambda seq: ''.join('  % .4e' % v for v in seq)
    if value.ndim == 1:
        return fmt(value)
    if value.ndim == 2:
        return ''.join(fmt(row) + '\n' for row in value)
    return ''


class Outputs:

    def __init__(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir = out_dir
        self.ds = self.model = self.model_path = self.trainer = None
        self.last_rmse = None

    # ---- wiring (run/template.py:58-64)
    def set_ds(self, ds):
        self.ds = ds

    def set_model(self, model, model_dir):
        self.model, self.model_path = model, model_dir + '/best.ckpt'

    def set_trainer(self, trainer):
        self.trainer = trainer

    def get_last_rmse(self):
        return self.last_rmse

    def _path(self, name):
        return self.out_dir + '/' + name

    def create_all(self):
        if self.model is None or self.ds is None:
            raise AssertionError('set_model() and set_ds() come first')
        with self.model.graph.as_default(), Session() as sess:
            self.model.saver.restore(sess, self.model_path)
            print("Generating outputs...")
            self._create_all(sess)

    def _create_all(self, sess):
        for step in (lambda: self.training_stats(), lambda: self.prediction(sess), lambda: self.test_mse(sess),
                     lambda: self.var_dump(sess)):
            step()

    # ---- training curve
    def training_stats(self):
        if self.trainer is None or plt is None or not is_writer():
            return
        print("  training stats")
        fig = plt.figure(1)
        for series, label in ((self.trainer.train_all, 'train'), (self.trainer.test_all, 'test')):
            plt.plot(series, label=label)
        plt.legend()
        fig.savefig(self._path('training_loss.pdf'))
        plt.close(fig)

    # ---- free-running prediction of one experiment
    def _free_run(self, sess, data_in, data_out, fetches):
        m = self.model
        m.load_ds(sess, data_in, data_out)
        return sess.run(fetches, feed_dict={m.condition: False})

    def _predict_one(self, sess, data_in, data_out, tag, steps):
        ds = self.ds
        first_in, first_out = data_in[:1, :steps], data_out[:1, :steps]
        mean_n, var_n = self._free_run(sess, first_in, first_out, (self.model.pred_mean, self.model.pred_var))
        mean = ds.denormalize(mean_n, 'out')[0]
        std = ds.denormalize(np.sqrt(var_n), 'out', shift=False)[0]
        truth = ds.denormalize(first_out, 'out')[0]
        if not is_writer():          # data parallel: every rank computes the same prediction, rank 0 writes the files
            return
        savemat(self._path('predict_%s.mat' % tag), {'mean': mean, 'std': std, 'gt': truth})
        if plt is None:
            return
        fig = plt.figure(1, figsize=(6, 4))
        plt.plot(truth[:, 0], label='ground truth')
        plt.plot(mean[:, 0], label='prediction')
        half = 1.96 * std[:, 0]
        plt.fill_between(np.arange(steps), mean[:, 0] - half, mean[:, 0] + half, color=_BAND_COLOUR)
        plt.legend(loc=2)
        plt.grid(True)
        plt.xlabel("time (steps)")
        plt.xlim([0, steps])
        fig.savefig(self._path('predict_%s.pdf' % tag), bbox_inches='tight')
        plt.close(fig)

    def prediction(self, sess, predict_size=300):
        print("  prediction")
        for tag in ('train', 'test'):
            data_in, data_out = getattr(self.ds, tag + '_in'), getattr(self.ds, tag + '_out')
            self._predict_one(sess, data_in, data_out, tag, min(predict_size, data_in.shape[1]))

    # ---- test error
    def test_mse(self, sess):
        print("  test mse")
        m, ds = self.model, self.ds
        per_experiment = []
        n_exp = ds.test_in.shape[0]
        batched = None
        if hasattr(m, 'run_experiments') and not os.environ.get('CBFSSM_OUTPUTS_LOOP'):
            # the reference runs one B = 1 sess.run per test experiment (outputs.py:127-133); here all of them go through
            # the kernels in one launch, each with the noise its own run would have drawn
            batched = m.run_experiments(sess, m.pred_mean, ds.test_in, ds.test_out, {m.condition: False})
        for k in range(n_exp):
            if batched is not None:
                pred = batched[k]
            else:
                m.load_ds(sess, ds.test_in[k:k + 1], ds.test_out[k:k + 1])
                pred = m.run(sess, m.pred_mean, {m.condition: False})[0]
            err = ds.denormalize(ds.test_out[k:k + 1], 'out')[0] - ds.denormalize(pred, 'out')[0]
            per_experiment.append(np.mean(err * err))            # mean over time and output dims, uniform weights
        mse = float(np.mean(per_experiment))
        self.last_rmse = float(np.sqrt(mse))
        if not is_writer():
            return
        with open(self._path('mse.txt'), 'w') as fh:
            fh.write("MSE:  %f\nRMSE: %f\n" % (mse, self.last_rmse))

    # ---- parameters
    def var_dump(self, sess):
        print("  var dump")
        m = self.model
        if not is_writer():
            return
        with open(self._path('var_dump.txt'), 'w') as fh:
            for name, fetch in m.var_dict.items():
                value = np.asarray(sess.run(fetch, feed_dict={m.condition: False}))
                fh.write('%s:\n%s\n\n' % (name, _rows_as_text(value)))

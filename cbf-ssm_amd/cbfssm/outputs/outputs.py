"""Evaluation outputs with the reference's interface and files (cbfssm/outputs/outputs.py:11-164): restore best.ckpt,
300-step open-loop prediction (condition=False) of the first train/test experiment -> predict_*.mat (+ .pdf when
matplotlib is importable), per-experiment test RMSE on denormalised outputs -> mse.txt, var_dump.txt."""
import math
import os
import numpy as np
import scipy.io

from ..model.session import Session

try:
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
except Exception:            # pragma: no cover
    plt = None


class Outputs:

    def __init__(self, out_dir):
        self.out_dir = out_dir
        self.ds = None
        self.model = None
        self.model_path = None
        self.trainer = None
        self.last_rmse = None
        os.makedirs(self.out_dir, exist_ok=True)

    def set_ds(self, ds):
        self.ds = ds

    def set_model(self, model, model_dir):
        self.model = model
        self.model_path = model_dir + '/best.ckpt'

    def set_trainer(self, trainer):
        self.trainer = trainer

    def get_last_rmse(self):
        return self.last_rmse

    def create_all(self):
        assert self.model is not None
        assert self.ds is not None
        with self.model.graph.as_default():
            with Session() as sess:
                self.model.saver.restore(sess, self.model_path)
                print("Generating outputs...")
                self._create_all(sess)

    def _create_all(self, sess):
        self.training_stats()
        self.prediction(sess)
        self.test_mse(sess)
        self.var_dump(sess)

    def training_stats(self):
        if self.trainer is not None and plt is not None:
            print("  training stats")
            plt.figure(1)
            plt.plot(self.trainer.train_all, label='train')
            plt.plot(self.trainer.test_all, label='test')
            plt.legend()
            plt.savefig(self.out_dir + '/training_loss.pdf')
            plt.close(1)

    def _predict_one(self, sess, data_in, data_out, tag, predict_size):
        model, ds = self.model, self.ds
        model.load_ds(sess, data_in[0:1, :predict_size, :], data_out[0:1, :predict_size, :])
        pred, var = sess.run((model.pred_mean, model.pred_var), feed_dict={model.condition: False})
        pred = ds.denormalize(pred, 'out')[0, :, :]
        gt = ds.denormalize(data_out[0:1, :predict_size, :], 'out')[0, :, :]
        std = ds.denormalize(np.sqrt(var), 'out', shift=False)[0, :, :]
        if plt is not None:
            lower, upper = pred[:, 0] - 1.96 * std[:, 0], pred[:, 0] + 1.96 * std[:, 0]
            plt.figure(1, figsize=(6, 4))
            plt.plot(gt[:, 0], label='ground truth')
            plt.plot(pred[:, 0], label='prediction')
            plt.fill_between(range(predict_size), lower, upper, color=(1.0, 178. / 255., 110. / 255.))
            plt.legend(loc=2)
            plt.grid(True)
            plt.xlabel("time (steps)")
            plt.xlim([0, predict_size])
            plt.savefig(self.out_dir + '/predict_%s.pdf' % tag, bbox_inches='tight')
            plt.close(1)
        scipy.io.savemat(self.out_dir + '/predict_%s.mat' % tag, {'mean': pred, 'std': std, 'gt': gt})

    def prediction(self, sess, predict_size=300):
        print("  prediction")
        ds = self.ds
        predict_size = min(ds.train_in.shape[1], predict_size)
        self._predict_one(sess, ds.train_in, ds.train_out, 'train', predict_size)
        self._predict_one(sess, ds.test_in, ds.test_out, 'test', min(ds.test_in.shape[1], predict_size))

    def test_mse(self, sess):
        print("  test mse")
        model, ds = self.model, self.ds
        mse_all = []
        for i in range(ds.test_in.shape[0]):
            model.load_ds(sess, ds.test_in[i:i + 1, :, :], ds.test_out[i:i + 1, :, :])
            pred = model.run(sess, model.pred_mean, {model.condition: False})[0]
            pred = ds.denormalize(pred, 'out')[0]
            gt = ds.denormalize(ds.test_out[i:i + 1, :, :], 'out')[0]
            mse_all.append(np.mean(np.square(gt - pred)))        # sklearn mean_squared_error, uniform average
        mse_all = float(np.mean(np.asarray(mse_all)))
        rmse_all = math.sqrt(mse_all)
        with open(self.out_dir + '/mse.txt', 'w') as f:
            f.write("MSE:  %f\n" % mse_all)
            f.write("RMSE: %f\n" % rmse_all)
        self.last_rmse = rmse_all

    def var_dump(self, sess):
        print("  var dump")
        model = self.model
        with open(self.out_dir + '/var_dump.txt', 'w') as f:
            for name, variable in model.var_dict.items():
                value = np.asarray(sess.run(variable, feed_dict={model.condition: False}))
                f.write(name + ":\n")
                if value.ndim == 1:
                    for val in value:
                        f.write("  % .4e" % val)
                elif value.ndim == 2:
                    for row in value:
                        for val in row:
                            f.write("  % .4e" % val)
                        f.write('\n')
                f.write("\n\n")

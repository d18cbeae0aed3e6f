"""`Trainer(model, model_dir).train(ds, epochs, retrain=False)` -- the call shape of the reference's
cbfssm/training/trainer.py:10-63.  An epoch is one pass over the training windows fetching (train, loss) and one pass over
the test windows fetching loss, both conditioned; `best.ckpt` is written whenever the epoch-mean training loss reaches a
new minimum, `model.ckpt` after the last epoch; `retrain=True` continues from `model.ckpt` instead of initialising."""
import os
import numpy as np

from ..model.session import Session


class Trainer:

    def __init__(self, model, model_dir):
        self.model, self.model_dir = model, model_dir
        self.train_all, self.test_all = [], []      # epoch-mean losses, in order

    def _ckpt(self, name):
        return os.path.join(self.model_dir, name + '.ckpt').replace(os.sep, '/')

    def _mean_loss(self, sess, data_in, data_out, with_update):
        m = self.model
        m.load_ds(sess, data_in, data_out)
        fetches = (m.train, m.loss) if with_update else m.loss
        losses = m.run(sess, fetches, {m.condition: True})
        return float(np.mean(losses[1] if with_update else losses))

    def train(self, ds, epochs, retrain=False):
        print('\nTraining...\n')
        m = self.model
        with m.graph.as_default(), Session() as sess:
            if retrain:
                m.saver.restore(sess, self._ckpt('model'))
            else:
                sess.run(m.init)
            best = np.inf
            for epoch in range(epochs):
                tr = self._mean_loss(sess, ds.train_in_batch, ds.train_out_batch, with_update=True)
                te = self._mean_loss(sess, ds.test_in_batch, ds.test_out_batch, with_update=False)
                self.train_all.append(tr)
                self.test_all.append(te)
                print('[%04d]: Train %s, Test %s' % (epoch, tr, te))
                if tr < best:
                    best = tr
                    m.saver.save(sess, self._ckpt('best'))
            m.saver.save(sess, self._ckpt('model'))

"""Trainer with the reference's signature and epoch semantics (cbfssm/training/trainer.py:10-63): per epoch one
training pass (`train` + `loss` per mini-batch) and one test pass (`loss` only), both with condition=True; best.ckpt
whenever the epoch-mean training loss improves, model.ckpt at the end; retrain=True resumes from model.ckpt."""
import numpy as np

from ..model.session import Session

try:
    from tqdm import tqdm
except ImportError:          # pragma: no cover
    def tqdm(x):
        return x


class Trainer:

    def __init__(self, model, model_dir):
        self.model = model
        self.model_dir = model_dir
        self.train_all = []
        self.test_all = []

    def train(self, ds, epochs, retrain=False):
        print('\nTraining...\n')
        model = self.model
        with model.graph.as_default():
            with Session() as sess:
                if retrain:
                    model.saver.restore(sess, self.model_dir + '/model.ckpt')
                else:
                    sess.run(model.init)
                lowest_train = float('inf')
                for epoch in tqdm(range(epochs)):
                    model.load_ds(sess, ds.train_in_batch, ds.train_out_batch)
                    train_loss = model.run(sess, (model.train, model.loss), {model.condition: True})
                    train_loss = np.mean(train_loss[1])
                    model.load_ds(sess, ds.test_in_batch, ds.test_out_batch)
                    test_loss = model.run(sess, model.loss, {model.condition: True})
                    test_loss = np.mean(test_loss)
                    print('[{epoch:04}]: Train {train}, Test {test}'.format(epoch=epoch, train=train_loss,
                                                                            test=test_loss))
                    self.train_all.append(train_loss)
                    self.test_all.append(test_loss)
                    if train_loss < lowest_train:
                        model.saver.save(sess, self.model_dir + '/best.ckpt')
                        lowest_train = train_loss
                model.saver.save(sess, self.model_dir + '/model.ckpt')

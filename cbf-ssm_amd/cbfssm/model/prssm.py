"""cbfssm.model.PRSSM on the HIP path: the PR-SSM baseline the reference re-implements for comparison (reference
cbfssm/model/prssm.py:14-172).  It is the forward pass of CBFSSMHALF without any conditioning on the observations:
x_{t+1} = GP mean + residual + eps*sqrt(var) everywhere, ELBO = loss_factors[0] * loglik - KL_z (not divided by the
particle count; the KL prior is factorised without jitter), one shared kernel lengthscale, recognition model for x_0
in {'output', 'conv', 'rnn'}.  `model.condition` may be fed (the callers do) but is ignored, as in the reference."""
import numpy as np

from .cbfssm import CBFSSM, backward
from .cbfssmhalf import _glorot
from .session import Fetch


class PRSSM(CBFSSM):

    _noise_with_backward = False

    # ---- prssm.py:27-47
    def _setup_vars(self):
        c = self.config
        self.dim_u, self.dim_y, self.dim_x = c['ds'].dim_u, c['ds'].dim_y, c['dim_x']
        M, D = c['ind_pnt_num'], self.dim_x + self.dim_u
        rng = self._rng
        assert np.asarray(c['var_y']).shape == (self.dim_y,), "PRSSM: config['var_y'] needs ds.dim_y entries"
        init = {'zeta_pos': rng.uniform(-c['zeta_pos'], c['zeta_pos'], size=(M, D)),
                'zeta_mean': c['zeta_mean'] * rng.random((M, self.dim_x)),
                'zeta_var_unc': backward(c['zeta_var'] * np.ones((M, self.dim_x))),
                'variance_unc': backward(c['gp_var']),
                'lengthscales_unc': backward(c['gp_len']),          # RBF(gp_var, gp_len): one shared lengthscale (:40)
                'var_x_unc': backward(c['var_x']), 'var_y_unc': backward(c['var_y'])}
        recog = c['recog_model']
        assert recog in ('output', 'conv', 'rnn'), 'invalid config for recognition model'    # prssm.py:171
        n_in = self.dim_u + self.dim_y
        if recog == 'rnn':                                                                    # prssm.py:158-169
            H = 16
            init.update({'recog.gate_kernel': _glorot(rng, n_in + H, 2 * H), 'recog.gate_bias': np.ones(2 * H),
                         'recog.cand_kernel': _glorot(rng, n_in + H, H), 'recog.cand_bias': np.zeros(H),
                         'recog.dense_kernel': _glorot(rng, H, self.dim_x), 'recog.dense_bias': np.zeros(self.dim_x)})
        elif recog == 'conv':                                                                 # prssm.py:143-155
            flat = 5 * ((int(c['recog_len']) - 2) // 2)          # the reference hard-codes 35 (= recog_len 16)
            lim = np.sqrt(6.0 / (3 * n_in + 3 * 5))
            init.update({'recog.conv_kernel': rng.uniform(-lim, lim, size=(3, n_in, 5)), 'recog.conv_bias': np.zeros(5),
                         'recog.dense_kernel': _glorot(rng, flat, self.dim_x), 'recog.dense_bias': np.zeros(self.dim_x)})
        self._init_values = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in init.items()}
        names = {'process noise': ('var_x_unc', True), 'observation noise': ('var_y_unc', True),
                 'kernel lengthscales': ('lengthscales_unc', True), 'kernel variance': ('variance_unc', True),
                 'IP pos': ('zeta_pos', False), 'IP mean': ('zeta_mean', False), 'IP var': ('zeta_var_unc', True)}
        self._var_spec = names
        self.var_dict = {k: Fetch(self, 'var:' + k) for k in names}                          # prssm.py:41-47

    def _make_engine(self, sess, dist):
        from ..hip.train_half import HipHalfGrad, half_param_names
        return HipHalfGrad(self.config, sess.device, dist, variant='prssm', dtype=self.dtype), half_param_names(self.config, 'prssm')

    def _execute(self, sess, names, feed, lazy=False):
        feed = dict(feed)
        feed.setdefault('condition', False)      # the PR-SSM graph has no use for the placeholder (prssm.py:19-130)
        return super()._execute(sess, names, feed, lazy=lazy)

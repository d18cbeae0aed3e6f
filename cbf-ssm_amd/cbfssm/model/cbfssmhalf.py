"""cbfssm.model.CBFSSMHALF on the HIP path: the forward-only variant the reference recommends for systems without
unstable hidden dimensions (run/template.py:17; reference cbfssm/model/cbfssmhalf.py:7-211).  Same attribute surface
as CBFSSM; config['var_y'] has ds.dim_y entries and config may carry 'recog_model' in {'rnn', 'output'}."""
import numpy as np
import torch

from .cbfssm import CBFSSM, backward
from .session import Fetch


def _glorot(rng, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))           # tf glorot_uniform, the default variable initializer
    return rng.uniform(-lim, lim, size=(fan_in, fan_out))


class CBFSSMHALF(CBFSSM):

    # ---- cbfssmhalf.py:20-47
    def _setup_vars(self):
        c = self.config
        self.dim_u, self.dim_y, self.dim_x = c['ds'].dim_u, c['ds'].dim_y, c['dim_x']
        M, D = c['ind_pnt_num'], self.dim_x + self.dim_u
        rng = self._rng
        init = {'f.zeta_pos': rng.uniform(-c['zeta_pos'], c['zeta_pos'], size=(M, D)),
                'f.zeta_mean': c['zeta_mean'] * rng.random((M, self.dim_x)),
                'f.zeta_var_unc': backward(c['zeta_var'] * np.ones((M, self.dim_x))),
                'f.variance_unc': backward(c['gp_var']),
                'f.lengthscales_unc': backward(np.asarray([c['gp_len']] * D, dtype=np.float64)),
                'var_x_unc': backward(c['var_x']), 'var_y_unc': backward(c['var_y'])}
        assert np.asarray(c['var_y']).shape == (self.dim_y,), "CBFSSMHALF: config['var_y'] needs ds.dim_y entries"
        recog = c.get('recog_model', 'rnn')
        assert recog in ('rnn', 'output'), 'invalid config for recognition model'            # cbfssmhalf.py:94
        if recog == 'rnn':                                                                    # cbfssmhalf.py:82-93
            n_in, H = self.dim_u + self.dim_y, 16
            init.update({'recog.gate_kernel': _glorot(rng, n_in + H, 2 * H), 'recog.gate_bias': np.ones(2 * H),
                         'recog.cand_kernel': _glorot(rng, n_in + H, H), 'recog.cand_bias': np.zeros(H),
                         'recog.dense_kernel': _glorot(rng, H, self.dim_x), 'recog.dense_bias': np.zeros(self.dim_x)})
        self._init_values = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in init.items()}
        names = {'process noise': ('var_x_unc', True), 'observation noise': ('var_y_unc', True),
                 'kernel lengthscales f': ('f.lengthscales_unc', True), 'kernel variance f': ('f.variance_unc', True),
                 'IP pos f': ('f.zeta_pos', False), 'IP mean f': ('f.zeta_mean', False),
                 'IP var f': ('f.zeta_var_unc', True)}
        self._var_spec = names
        self.var_dict = {k: Fetch(self, 'var:' + k) for k in names}                          # cbfssmhalf.py:41-47

    def _make_engine(self, sess, dist):
        from ..hip.train_half import HipHalfGrad, half_param_names
        return HipHalfGrad(self.config, sess.device, dist, dtype=self.dtype), half_param_names(self.config)

    _noise_with_backward = False

"""cbfssm.model.CBFSSM on the MI355X HIP path: the attribute surface the reference's callers use
(cbfssm/model/cbfssm.py:10-277 as consumed by training/trainer.py:18-63 and outputs/outputs.py:36-164):

    CBFSSM(config)            config dict of run/template.py:19-40; config['ds'] is a class with dim_u/dim_y
    .graph .init .saver .condition .train .loss .pred_mean .pred_var .internal_mean .internal_var .mse .sde .var_dict
    .load_ds(sess, in, out)   .run(sess, tensors, feed_dict)

One `sess.run` evaluates one mini-batch: noise is drawn on the device (one normal per (b, s) and step, tiled over the
state dimensions, cbfssm.py:134,149,209), the ELBO and -- when `train` is fetched -- its gradient and the Adam update
run in the hand-written kernels behind include/cbfssm_hip.h.  There is no CPU path.
"""
import os
import numpy as np
import torch

from .base_model import BaseModel
from .session import Fetch, Saver, InvalidArgumentError, OutOfRangeError
from ..synthetic import softplus_inverse

_FETCHES = ('train', 'loss', 'pred_mean', 'pred_var', 'internal_mean', 'internal_var', 'mse', 'sde',
            'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b', 'x_final', 'y_tilde')


def backward(y):
    """inverse softplus of the positive initial values (cbfssm/model/tf_transform.py:13-16)."""
    return softplus_inverse(y)


class CBFSSM(BaseModel):

    def __init__(self, config, dtype='float64'):
        """dtype: 'float64' (the reference default, cbfssm.py:12) or 'float32' (also accepted: torch / numpy / TensorFlow
        dtype objects of those names).  A float32 model runs its time loops in float32 with the Cholesky kept in
        float64 (gp_tf.py:57-65) and serves loss / prediction fetches; `model.train` needs float64."""
        name = str(dtype)
        if 'float64' in name or name in ('double', 'f64'):
            dt = 'float64'
        elif 'float32' in name or name in ('float', 'f32'):
            dt = 'float32'
        else:
            raise NotImplementedError('dtype %r: the HIP path computes in float64 or float32' % (dtype,))
        super(CBFSSM, self).__init__(config, dtype=dt)

    # ---- cbfssm.py:15-23
    def _build_graph(self):
        self._setup_vars()
        for name in _FETCHES:
            setattr(self, name, Fetch(self, name))
        self.init = Fetch(self, 'init')
        self.saver = Saver(self)
        self._engine = None
        self._opt = None
        self._device = None
        self._gen = None
        self._dist = None

    # ---- cbfssm.py:25-67 (numpy initial values; unseeded unless config['seed'] is given)
    def _setup_vars(self):
        c = self.config
        self.dim_u, self.dim_y, self.dim_x = c['ds'].dim_u, c['ds'].dim_y, c['dim_x']
        M, D = c['ind_pnt_num'], self.dim_x + self.dim_u
        rng = self._rng
        init = {}
        for g, dout in (('f', self.dim_x), ('b', self.dim_x - self.dim_y)):
            init[g + '.zeta_pos'] = rng.uniform(-c['zeta_pos'], c['zeta_pos'], size=(M, D))          # gp_tf.py:112-115
            init[g + '.zeta_mean'] = c['zeta_mean'] * rng.random((M, dout))                           # gp_tf.py:117-118
            init[g + '.zeta_var_unc'] = backward(c['zeta_var'] * np.ones((M, dout)))                  # gp_tf.py:120-121
            init[g + '.variance_unc'] = backward(c['gp_var'])                                         # gp_tf.py:25-26
            init[g + '.lengthscales_unc'] = backward(np.asarray([c['gp_len']] * D, dtype=np.float64))  # gp_tf.py:29-30
        init['var_x_unc'] = backward(c['var_x'])                                                      # cbfssm.py:51
        init['var_y_unc'] = backward(c['var_y'])                                                      # cbfssm.py:53
        self._init_values = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in init.items()}
        names = {'process noise': ('var_x_unc', True), 'observation noise': ('var_y_unc', True)}
        for g in 'fb':
            names['kernel lengthscales ' + g] = (g + '.lengthscales_unc', True)
            names['kernel variance ' + g] = (g + '.variance_unc', True)
            names['IP pos ' + g] = (g + '.zeta_pos', False)
            names['IP mean ' + g] = (g + '.zeta_mean', False)
            names['IP var ' + g] = (g + '.zeta_var_unc', True)
        self._var_spec = names
        self.var_dict = {k: Fetch(self, 'var:' + k) for k in names}                                   # cbfssm.py:56-67

    # ---- engine plumbing
    def _ensure(self, sess):
        if self._engine is not None:
            return
        from ..hip.train import TFAdam
        from ..hip.dist_utils import active
        dist = active()
        self._dist = dist
        self._device = sess.device
        self._engine, names = self._make_engine(sess, dist)
        params = {k: torch.tensor(self._init_values[k], device=sess.device) for k in names}
        self._opt = TFAdam(params, self.config['learning_rate'])                                      # cbfssm.py:274
        self._gen = torch.Generator(device=sess.device)
        seed = self.config.get('seed', None)
        seed = int(seed) if seed is not None else int.from_bytes(os.urandom(4), 'little')
        if dist is not None:
            # data parallel: same iteration order on every rank, each rank takes its shard of every mini-batch
            from ..hip.dist_utils import broadcast_seed, broadcast_tensor
            seed = broadcast_seed(seed, dist)
            self._rng = np.random.default_rng(seed)
            self._rank, self._world = dist.get_rank(), dist.get_world_size()
            self._gen.manual_seed(seed + 7919 * (self._rank + 1))
            # evaluations that are not sharded (predictions: every rank computes the whole batch) draw the same noise
            # on every rank
            self._gen_common = torch.Generator(device=sess.device)
            self._gen_common.manual_seed(seed)
            # the reference draws its initial inducing inputs / means unseeded (gp_tf.py:112-118): every rank has drawn
            # its own in _setup_vars.  Rank 0's values win, also for a later `sess.run(model.init)`.
            broadcast_tensor(self._opt.flat, dist, src=0)
            self._init_values = {k: self._opt.views[k].detach().cpu().numpy().copy() for k in names}
        else:
            self._gen.manual_seed(seed)

    def _make_engine(self, sess, dist):
        from ..hip.train import HipElboGrad, PARAM_NAMES
        return HipElboGrad(self.config, sess.device, dist, require_adjoint=False, dtype=self.dtype), PARAM_NAMES

    def _state_dict(self):
        sd = self._opt.state_dict()
        return {'flat': sd['flat'].cpu(), 'm': sd['m'].cpu(), 'v': sd['v'].cpu(), 't': int(sd['t']),
                'names': list(sd['names'])}

    def _load_state_dict(self, sess, sd):
        self._ensure(sess)
        assert list(sd['names']) == list(self._opt.names), 'checkpoint belongs to a different model'
        dev = self._device
        self._opt.load_state_dict({'flat': sd['flat'].to(dev), 'm': sd['m'].to(dev), 'v': sd['v'].to(dev), 't': sd['t']})

    _noise_with_backward = True

    # ---- mini-batches: host gather + upload of batch k+1 overlap the device work of batch k
    _END = object()

    def _stage_batch(self):
        data_in, data_out = self._next_batch()                      # raises OutOfRangeError at the end of the data
        # (data parallel: the whole mini-batch is staged on every rank -- 7 MB at C3 -- and sharded on the device in
        #  _execute, where the fetch list says whether this run is sharded at all)
        if getattr(self, '_upload', None) is None:
            self._upload = torch.cuda.Stream(device=self._device)
        # through page-locked staging buffers (two per shape, alternating: the copy of batch k+1 is asynchronous while
        # batch k's buffer may still be in flight); a pageable upload of fresh numpy pages every step is at the mercy
        # of the host's paging state (observed: sporadically 20 ms instead of 0.7 ms for 7 MB)
        pins = getattr(self, '_pins', None)
        if pins is None:
            pins = self._pins = {}
        key = (data_in.shape, data_out.shape)
        if key not in pins:
            pins[key] = [[torch.empty(s, dtype=torch.float64).pin_memory() for s in key] + [None] for _ in range(2)] + [0]
        slot = pins[key][pins[key][2]]
        pins[key][2] ^= 1
        if slot[2] is not None:
            slot[2].synchronize()         # this slot's previous upload (two batches ago; BaseModel.run lets the host run ahead)
        slot[0].numpy()[...] = data_in
        slot[1].numpy()[...] = data_out
        with torch.cuda.stream(self._upload):                       # not behind the kernels of the running step
            u = slot[0].to(self._device, non_blocking=True)
            y = slot[1].to(self._device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._upload)
        slot[2] = ev
        return u, y, ev

    def _stage_ahead(self):
        if self._staged is None:
            try:
                self._staged = self._stage_batch()
            except OutOfRangeError:
                self._staged = self._END

    def _device_batch(self):
        staged, self._staged = self._staged, None
        if staged is self._END:
            raise OutOfRangeError()
        u, y, ev = staged if staged is not None else self._stage_batch()
        cur = torch.cuda.current_stream(self._device)
        cur.wait_event(ev)
        u.record_stream(cur)
        y.record_stream(cur)
        return u, y

    def _train_stepper(self):
        """The HIP-graph stepper over this model's engine and optimiser: HipTrainStep for CBFSSM, HipHalfTrainStep for the
        forward-only variants (their recognition network's autograd launches are captured too)."""
        from ..hip.train import HipElboGrad, HipTrainStep
        from ..hip.train_half import HipHalfGrad, HipHalfTrainStep
        if getattr(self, '_stepper', None) is None or self._stepper.engine is not self._engine:
            if type(self._engine) is HipElboGrad:
                self._stepper = HipTrainStep(self.config, None, self._device, self._dist, engine=self._engine, opt=self._opt)
            elif type(self._engine) is HipHalfGrad:
                self._stepper = HipHalfTrainStep(self._engine, self._opt)
            else:
                return None
        return self._stepper if self._stepper.use_graph else None

    def _draw_noise(self, B, T, common=False):
        from ..hip.ops import NoisePipeline
        if common:
            if getattr(self, '_noise_common', None) is None:
                self._noise_common = NoisePipeline(self._device, self._gen_common, self._noise_with_backward)
            return self._noise_common.next(T, B * self.config['samples'])
        if getattr(self, '_noise', None) is None:
            self._noise = NoisePipeline(self._device, self._gen, self._noise_with_backward)
        return self._noise.next(T, B * self.config['samples'])

    def run_experiments(self, sess, fetch, data_in, data_out, feed_dict):
        """What `for k: load_ds(in[k:k+1], out[k:k+1]); run(sess, fetch, feed)` returns (outputs/outputs.py:121-133 of the
        reference loops over the test experiments that way, one B = 1 `sess.run` each: 2 workgroups on a 256-CU part), as
        ONE launch over all experiments: they are rows of one array, hence of equal length, and sequences never
        interact.  Every experiment gets the noise the k-th run of that loop would have drawn (same generator, same
        order), so the results are the loop's results.  Returns a list with one array per experiment."""
        self._ensure(sess)
        name = fetch.name
        if name not in ('pred_mean', 'pred_var', 'internal_mean', 'internal_var'):
            raise KeyError('run_experiments serves the per-sequence prediction fetches, not %r' % (name,))
        feed = {(k.name if hasattr(k, 'name') else k): v for k, v in feed_dict.items()}
        condition = bool(feed['condition'])
        data_in = np.ascontiguousarray(data_in, dtype=np.float64)
        data_out = np.ascontiguousarray(data_out, dtype=np.float64)
        n, T = data_in.shape[0], data_in.shape[1]
        S = self.config['samples']
        common = self._dist is not None
        draws = [{k: v.clone() for k, v in self._draw_noise(1, T, common=common).items()} for _ in range(n)]
        noise = {}
        for k in draws[0]:
            lead = 2 if k in ('hid_b', 'eps_b') else 1
            noise[k] = torch.cat([d[k].view(lead, -1, S) for d in draws], dim=2).contiguous()     # chain c = b S + s
        u = torch.as_tensor(data_in, device=self._device)
        y = torch.as_tensor(data_out, device=self._device)
        kw = {'local': True} if self._dist is not None else {}
        loss, terms, ws = self._engine.forward(self._opt.views, u, y, noise, condition, **kw)
        if float(terms['info']) != 0.0:
            raise InvalidArgumentError('Cholesky decomposition was not successful')
        res = getattr(ws, {'internal_mean': 'int_mean', 'internal_var': 'int_var'}.get(name, name)).cpu().numpy()
        return [res[k:k + 1] for k in range(n)]

    _SHARDED_FETCHES = frozenset(('train', 'loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b'))

    # ---- one sess.run
    _SCALAR_FETCHES = ('train', 'loss', 'loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b')

    def _execute(self, sess, names, feed, lazy=False):
        """One `sess.run`.  lazy (BaseModel.run, scalar fetches only): returns the device row [info, loss, scalars...] instead
        of host values -- the caller reads all rows of a pass over the data back at once, so that no step waits for the host."""
        self._ensure(sess)
        if names == ['init']:
            for k, v in self._init_values.items():
                self._opt.views[k].copy_(torch.tensor(v, device=self._device))
            self._opt.m.zero_()
            self._opt.v.zero_()
            self._opt.t = 0
            return [None]
        if all(n.startswith('var:') for n in names):
            return [self._var_value(n[4:]) for n in names]
        if 'condition' not in feed:
            raise KeyError('feed_dict must set model.condition (cbfssm.py:227)')
        condition = bool(feed['condition'])
        u, y = self._device_batch()
        # Data parallel: runs that fetch only the loss (and `train`) shard the mini-batch over the ranks and exchange one
        # all-reduce; runs that fetch per-sequence results (pred_mean, x_final, ...) are evaluated whole on every rank,
        # without a collective and with rank-invariant noise, so every rank holds the same result.  A rank whose shard is
        # empty (fewer sequences than ranks: run_sarcos.py has batch_size 5, a partial last batch) evaluates a one-
        # sequence stand-in with weight 0 and still joins the collective.
        kw = {}
        sharded = self._dist is not None and all(n in self._SHARDED_FETCHES for n in names)
        if sharded:
            from ..hip.dist_utils import shard_range
            lo, hi = shard_range(u.shape[0], self._rank, self._world)
            if hi <= lo:
                lo, hi, kw = 0, 1, {'weight': 0.0}
            u, y = u[lo:hi].contiguous(), y[lo:hi].contiguous()
        elif self._dist is not None:
            kw = {'local': True}
        B, T = u.shape[0], u.shape[1]
        noise = self._draw_noise(B, T, common=(self._dist is not None and not sharded))
        eng = self._engine
        if 'train' in names:
            stepper = self._train_stepper()
            if stepper is not None:                       # loss, gradient and Adam update as HIP graph replays
                loss = stepper.step(u, y, noise, condition, **kw)
                terms = stepper.last_terms
                ws = stepper.last_ws
            else:
                loss, grads, terms = eng.loss_and_grads(self._opt.views, u, y, noise, condition, **kw)
                self._opt.step(grads)                                                                 # cbfssm.py:275
                ws = eng.last_ws
        else:
            loss, terms, ws = eng.forward(self._opt.views, u, y, noise, condition, **kw)
        self._stage_ahead()     # the next mini-batch: gathered and uploaded while the device works on this one
        # one device-to-host transfer for everything scalar that this run fetches (each .item() is a stream sync)
        scal_names = [k for k in ('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b') if k in names]
        dev0 = dict(dtype=torch.float64, device=self._device)
        row = torch.stack([torch.as_tensor(v, **dev0).reshape(())
                           for v in [terms['info'], loss] + [terms[k] for k in scal_names]])
        if lazy:
            assert all(n in self._SCALAR_FETCHES for n in names)
            return row
        host = row.cpu().numpy()
        return self._scalar_results(names, host, ws, y, B, T)

    def _scalar_results(self, names, host, ws=None, y=None, B=None, T=None):
        """host = [info, loss, scalars in _SCALAR_FETCHES order...] of one run -> the fetched values"""
        scal_names = [k for k in ('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b') if k in names]
        info, loss_h = float(host[0]), float(host[1])
        scal_h = {k: float(host[2 + i]) for i, k in enumerate(scal_names)}
        if info != 0.0:
            raise InvalidArgumentError('Cholesky decomposition was not successful: leading minor %d of K_mm + 1e-8 I '
                                       'is not positive definite' % int(info))
        S = self.config['samples']
        out = []
        for n in names:
            if n == 'train':
                out.append(None)
            elif n == 'loss':
                out.append(np.float64(loss_h))
            elif n in ('loglik', 'kl_x', 'entropy', 'kl_z_f', 'kl_z_b'):
                out.append(np.float64(scal_h[n]))
            elif n in ('pred_mean', 'pred_var', 'internal_mean', 'internal_var'):
                out.append(getattr(ws, {'internal_mean': 'int_mean', 'internal_var': 'int_var'}.get(n, n)).cpu().numpy())
            elif n == 'mse':                                                                          # cbfssm.py:270
                out.append(np.float64(float(torch.mean((ws.pred_mean - y) ** 2))))
            elif n == 'sde':                                                                          # cbfssm.py:271
                out.append((torch.abs(ws.pred_mean - y) / torch.sqrt(ws.pred_var)).cpu().numpy())
            elif n == 'x_final':                                                                      # cbfssm.py:181
                out.append(ws.x.view(T, B, S, self.dim_x).permute(1, 0, 2, 3).cpu().numpy())
            elif n == 'y_tilde':                                                                      # cbfssm.py:95-97
                y2 = ws.y2.view(T, B, S, self.dim_x - self.dim_y).permute(1, 0, 2, 3)
                out.append(torch.cat((y[:, :, None, :].expand(B, T, S, self.dim_y), y2), dim=3).cpu().numpy())
            else:
                raise KeyError(n)
        return out

    def _var_value(self, name):
        from ..hip.ops import tf_forward
        key, positive = self._var_spec[name]
        v = self._opt.views[key]
        return (tf_forward(v) if positive else v).cpu().numpy()

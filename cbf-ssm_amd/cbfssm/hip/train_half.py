"""CBFSSMHALF on the HIP path (reference cbfssm/model/cbfssmhalf.py): forward-only variant -- x_0 comes from a
recognition model, the Kalman-style update acts on the observed state dims only, the ELBO has no backward GP and no
entropy term (cbfssmhalf.py:174-199).

The time loop and its adjoint run in the same kernels as CBFSSM's forward pass (problem.half = 1,
include/cbfssm_hip.h: cbfssm_half_forward_pass_f64 / _bwd_f64).  The recognition model is a GRU(16) over
recog_len steps + a dense layer on B sequences (cbfssmhalf.py:82-93) -- a few hundred FLOPs per sequence; it stays in
PyTorch autograd (float64, TF-1.8 GRUCell gate layout) and receives d loss / d x_0 from the adjoint kernel.
"""
import ctypes as C
import os
import math
import torch

from . import lib as _l
from . import ops
from .ops import _ptr, _stream, _f64, tf_forward, GPPack
from .dist_utils import all_reduce_sum
from .train import HipElboGrad, LOG2PI, StashContract, FlatDict

GP_NAMES = ('f.zeta_pos', 'f.zeta_mean', 'f.zeta_var_unc', 'f.variance_unc', 'f.lengthscales_unc')
RECOG_NAMES = ('recog.gate_kernel', 'recog.gate_bias', 'recog.cand_kernel', 'recog.cand_bias', 'recog.dense_kernel',
               'recog.dense_bias')
GRU_UNITS = 16      # cbfssmhalf.py:84


CONV_NAMES = ('recog.conv_kernel', 'recog.conv_bias', 'recog.dense_kernel', 'recog.dense_bias')
PRSSM_GP_NAMES = ('zeta_pos', 'zeta_mean', 'zeta_var_unc', 'variance_unc', 'lengthscales_unc')


def half_param_names(config, variant='half'):
    if variant == 'prssm':
        names = PRSSM_GP_NAMES + ('var_x_unc', 'var_y_unc')
        recog = config['recog_model']
    else:
        names = GP_NAMES + ('var_x_unc', 'var_y_unc')
        recog = config.get('recog_model', 'rnn')
    if recog == 'rnn':
        names = names + RECOG_NAMES
    elif recog == 'conv':
        names = names + CONV_NAMES
    return names


def conv_recognition(recog, u, y, recog_len):
    """prssm.py:143-155: conv1d(5, 3, relu) -> max_pool(2, 2) -> dense, in float32 as the reference casts it."""
    uy = torch.cat((u, y), dim=2)[:, :recog_len, :].to(torch.float32)
    k = recog['recog.conv_kernel'].to(torch.float32)                      # TF layout (width, in, out)
    x = torch.nn.functional.conv1d(uy.permute(0, 2, 1), k.permute(2, 1, 0), recog['recog.conv_bias'].to(torch.float32))
    x = torch.nn.functional.max_pool1d(torch.relu(x), 2, 2)
    x = x.permute(0, 2, 1).reshape(u.shape[0], -1)
    out = x @ recog['recog.dense_kernel'].to(torch.float32) + recog['recog.dense_bias'].to(torch.float32)
    return out.to(torch.float64)


def gru_recognition(recog, u, y, recog_len):
    """TF-1.8 GRUCell(16) over the reversed first recog_len steps of [u, y], then dense -> dim_x (cbfssmhalf.py:82-93)."""
    uy = torch.flip(torch.cat((u, y), dim=2)[:, :recog_len, :], dims=[1])
    h = torch.zeros(u.shape[0], GRU_UNITS, dtype=u.dtype, device=u.device)
    for t in range(uy.shape[1]):
        x = uy[:, t, :]
        gates = torch.sigmoid(torch.cat((x, h), 1) @ recog['recog.gate_kernel'] + recog['recog.gate_bias'])
        r, z = torch.chunk(gates, 2, dim=1)
        c = torch.tanh(torch.cat((x, r * h), 1) @ recog['recog.cand_kernel'] + recog['recog.cand_bias'])
        h = z * h + (1.0 - z) * c
    return h @ recog['recog.dense_kernel'] + recog['recog.dense_bias']


class HipHalfGrad:
    """loss and gradients of CBFSSMHALF for one mini-batch on one device."""

    def __init__(self, config, device, dist=None, variant='half', dtype='float64'):
        """variant 'half': CBFSSMHALF.  variant 'prssm': the PR-SSM baseline (reference cbfssm/model/prssm.py) -- the same
        pass with the Kalman update switched off everywhere (free run), loss = -(lambda0 * loglik - KL_z) with the KL
        prior factorised WITHOUT jitter (prssm.py:81-82,96) and one shared lengthscale (prssm.py:40).

        dtype 'float32' (the reference's model dtype argument, cbfssmhalf.py:17 / prssm.py:17): the time loop and its adjoint
        compute in float32 (cbfssm_half_forward_pass_f32 / _bwd_f32); K_mm / Cholesky / K^-1, the recognition model, the train
        tail and the optimizer step stay float64, as in HipElboGrad."""
        assert dtype in ('float64', 'float32')
        self.f32 = dtype == 'float32'
        self.config = config
        self.variant = variant
        self.device = torch.device(device)
        self.dist = dist
        self.dim_u, self.dim_y, self.dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
        self.M, self.S = config['ind_pnt_num'], config['samples']
        self.D = self.dim_x + self.dim_u
        self.names = half_param_names(config, variant)
        self.rnn = 'recog.gate_kernel' in self.names
        self.conv = 'recog.conv_kernel' in self.names
        self.pre = '' if variant == 'prssm' else 'f.'
        # loss = -(cL * (loglik - kl_x) - KL_z): cL = lambda0 / S for CBFSSMHALF, lambda0 for PR-SSM
        lf0 = float(config['loss_factors'][0])
        self.cL = lf0 if variant == 'prssm' else lf0 / self.S
        self.pack_kl = GPPack(self.M, self.D, self.dim_x, self.device) if variant == 'prssm' else None
        self.pack_f = GPPack(self.M, self.D, self.dim_x, self.device, ops.gp_form_mode_f32(config) if self.f32 else None)
        if self.f32:
            self.pack_f.cond_threshold = min(self.pack_f.cond_threshold, ops.F32_FORM_COND)
            # per-workgroup slabs of the float32 adjoint: the non-stash layout at every tile height (matrix section included)
            self.slab32_f = int(_l.load().cbfssm_rev32_slab_elems(C.byref(self.pack_f.layout)))
        self.stash = bool(self.pack_f.layout.rev_stash)
        self.stash_bytes = int(float(config.get('adjoint_stash_gib', 4.0)) * 2 ** 30)
        self._stash_buf = None
        self.slab_f = int(self.pack_f.layout.rev_slab)
        self._ws = {}
        self.last_ws = None
        # the K_mm / Cholesky / prior-KL adjoint is shared with CBFSSM: cbfssm_train_tail_half_f64 (five launches on flat
        # vectors); CBFSSM_TORCH_TAIL=1 keeps the tensor-library restatement (same numbers, a cross-check)
        self._gp_adjoint = HipElboGrad._gp_adjoint.__get__(self)
        self.fused_tail = self.slab_f > 0 and not os.environ.get('CBFSSM_TORCH_TAIL')
        self.gp_names = self.names[:7]                       # the five GP tensors, var_x_unc, var_y_unc: the tail's flat order
        self.tail_work = None
        # the GRU recognition model as two launches (cbfssm_gru_recog[_bwd]_f64: one wave per sequence) instead of ~800 tensor-
        # library launches through autograd; CBFSSM_TORCH_GRU=1 keeps the latter (same numbers, a cross-check)
        self.fused_gru = self.rnn and not os.environ.get('CBFSSM_TORCH_GRU')
        self._gru = {}

    def _timed(self, kind, fn):
        """measurement hook (bench.py --model half|prssm): with a list in self._prof the launch is bracketed by HIP events"""
        prof = getattr(self, '_prof', None)
        if prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        prof.append((kind, e0, e1))
        return rc

    def _problem(self, B, T, condition):
        c = self.config
        if self.variant == 'prssm':      # never condition: recog_len 1 and condition False make every step a free run
            return _l.make_problem(B, self.S, T, self.dim_x, self.dim_u, self.dim_y, self.M, 1, 1.0, False, half=True)
        return _l.make_problem(B, self.S, T, self.dim_x, self.dim_u, self.dim_y, self.M, c['recog_len'], c['k_factor'],
                               condition, half=True)

    def _recog_params(self, p):
        if self.rnn:
            return RECOG_NAMES
        return CONV_NAMES if self.conv else ()

    def _recog(self, rp, u, y):
        if self.rnn:
            return gru_recognition(rp, u, y, self.config['recog_len'])
        return conv_recognition(rp, u, y, self.config['recog_len'])

    def _recog_flat(self, params, p):
        """the six recognition tensors as one flat vector: the tail of the optimiser's own storage when `params` are its views"""
        flat = getattr(params, 'flat', None)
        n = sum(p[k].numel() for k in RECOG_NAMES)
        if flat is not None and flat.device == self.device and tuple(params.keys())[-6:] == RECOG_NAMES:
            return flat[flat.numel() - n:]
        return torch.cat([p[k].reshape(-1) for k in RECOG_NAMES])

    def _gru_forward(self, rflat, u, y, keep):
        lib = _l.load()
        B, T = u.shape[0], u.shape[1]
        R = min(int(self.config['recog_len']), T)            # (the window is the first recog_len steps: all of a shorter sequence)
        f = dict(dtype=torch.float64, device=self.device)
        key = (B, R)
        if key not in self._gru:
            P = int(lib.cbfssm_gru_recog_param_elems(self.dim_u, self.dim_y, self.dim_x))
            self._gru[key] = {'x0': torch.zeros(B, self.dim_x, **f), 'P': P,
                              'act': torch.zeros(int(lib.cbfssm_gru_recog_act_elems(B, R)), **f),
                              'gpart': torch.zeros((B + 32) * P, **f)}
        g = self._gru[key]
        rc = lib.cbfssm_gru_recog_f64(B, T, self.dim_u, self.dim_y, self.dim_x, R, _ptr(u), _ptr(y), _ptr(rflat), _ptr(g['x0']),
                                      _ptr(g['act']) if keep else None, _stream())
        _l.check(rc, 'cbfssm_gru_recog_f64')
        return g

    def _x0(self, p, u, y, params=None):
        if self.fused_gru:
            return self._gru_forward(self._recog_flat(params, p), u, y, keep=False)['x0']
        if self.rnn or self.conv:
            return self._recog(p, u, y)
        B = u.shape[0]
        return torch.cat((y[:, 0, :], torch.zeros(B, self.dim_x - self.dim_y, dtype=u.dtype, device=u.device)), dim=1)

    def _forward(self, p, c, x0, u, y, eps_f, prob, ws):
        lib = _l.load()
        st = _stream()
        pb = C.byref(prob)
        pre = self.pre
        self.pack_f.prepare(p[pre + 'zeta_pos'], c['ls'], c['var'], p[pre + 'zeta_mean'], c['zvar'])
        if self.pack_kl is not None:
            self.pack_kl.prepare(p[pre + 'zeta_pos'], c['ls'], c['var'], p[pre + 'zeta_mean'], c['zvar'], jitter=0.0)
        lay = C.byref(self.pack_f.layout)
        if self.f32:
            b32 = C.c_void_p(self.pack_f.pack_f32().data_ptr())
            rc = self._timed('forward_pass', lambda: lib.cbfssm_half_forward_pass_f32(
                pb, lay, b32, _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y), _ptr(x0),
                _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), _ptr(ws.kl_part), st))
            _l.check(rc, 'cbfssm_half_forward_pass_f32')
        else:
            rc = self._timed('forward_pass', lambda: lib.cbfssm_half_forward_pass_f64(
                pb, lay, _ptr(self.pack_f.buf), _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y), _ptr(x0),
                _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), _ptr(ws.kl_part), st))
            _l.check(rc, 'cbfssm_half_forward_pass_f64')
        rc = lib.cbfssm_loglik_moments_f64(pb, _ptr(c['var_y']), _ptr(y), _ptr(ws.x), _ptr(ws.ll_part),
                                           _ptr(ws.pred_mean), _ptr(ws.pred_var), _ptr(ws.int_mean), _ptr(ws.int_var), st)
        _l.check(rc, 'cbfssm_loglik_moments_f64')
        klp = self.pack_kl if self.pack_kl is not None else self.pack_f
        rc = lib.cbfssm_elbo_combine_f64(pb, self.cL * self.S, 0.0, _ptr(ws.ll_part), ws.ll_part.numel(), _ptr(ws.kl_part),
                                         ws.kl_part.numel(), None, 0, _ptr(klp.scal), None, _ptr(ws.out), st)
        _l.check(rc, 'cbfssm_elbo_combine_f64')

    def _constrained(self, p):
        pre = self.pre
        ls = tf_forward(p[pre + 'lengthscales_unc']).reshape(-1)
        if ls.numel() == 1:                                  # PR-SSM: one lengthscale for all input dims (prssm.py:40)
            ls = ls.expand(self.D)
        return {'ls': ls.contiguous(),
                'var': tf_forward(p[pre + 'variance_unc']).reshape(-1).contiguous(),
                'zvar': tf_forward(p[pre + 'zeta_var_unc']).contiguous(),
                'var_x': tf_forward(p['var_x_unc']).contiguous(), 'var_y': tf_forward(p['var_y_unc']).contiguous()}

    def _workspace(self, prob):
        key = (prob.B, prob.T)
        if key not in self._ws:
            lib = _l.load()
            f = dict(dtype=torch.float64, device=self.device)
            N = prob.B * prob.S

            class WS:
                pass
            ws = WS()
            ws.n_kl = int(lib.cbfssm_forward_pass_partials(C.byref(prob)))
            ws.n_f = int(lib.cbfssm_rev_workgroups(C.byref(prob), 0))
            ws.x = torch.zeros(prob.T, N, prob.dim_x, **f)
            ws.fmv_f = torch.zeros(max(prob.T - 1, 0), N, prob.dim_x, 2, **f)
            if self.f32:    # every step's [A2 | kernel tile] registers of the float32 pass, float32 (held in a float64 buffer)
                n_a2 = int(lib.cbfssm_saved_a2_f32_elems(C.byref(prob), C.byref(self.pack_f.layout), 0))
                assert n_a2 > 0 or prob.T == 1
                n_a2 = max((n_a2 + 1) // 2, 1)
            else:
                n_a2 = int(lib.cbfssm_saved_a2_elems(C.byref(prob), C.byref(self.pack_f.layout), 0))
            if getattr(self, 'tile_pool', None) is None:
                self.tile_pool = ops.TilePool(self.device)
            ws.a2s_f, _ = self.tile_pool.get(n_a2, 0)
            ws.kl_part = torch.zeros(ws.n_kl, **f)
            ws.ll_part = torch.zeros(int(lib.cbfssm_loglik_partials(C.byref(prob))), **f)   # [block][dim_y]
            ws.pred_mean = torch.zeros(prob.B, prob.T, prob.dim_y, **f)
            ws.pred_var = torch.zeros(prob.B, prob.T, prob.dim_y, **f)
            ws.int_mean = torch.zeros(prob.B, prob.T, prob.dim_x, **f)
            ws.int_var = torch.zeros(prob.B, prob.T, prob.dim_x, **f)
            ws.out = torch.zeros(8, **f)
            ws.y2 = torch.zeros(prob.T, N, max(0, prob.dim_x - prob.dim_y), **f)      # (surface compatibility)
            ws.gx0 = torch.zeros(N, prob.dim_x, **f)
            ws.gx_carry = torch.zeros(N, prob.dim_x, **f)
            ws.gpart_f = torch.zeros((ws.n_f + 32) * (self.slab32_f if self.f32 else self.slab_f), **f)
            ws.red = torch.zeros(self.slab_f + 3 + prob.dim_y, **f)     # [slab | loglik, kl_x, 0, d loss / d var_y]
            self._ws[key] = ws
        return self._ws[key]

    def _terms(self, ws, red2=None):
        out = ws.out
        cL = self.cL
        if red2 is None:
            loglik, kl_x = out[0], out[1]
        else:
            loglik, kl_x = red2[0], red2[1]
        loss = -(loglik * cL - kl_x * cL - out[3])                                 # cbfssmhalf.py:195-199
        z = torch.zeros((), dtype=torch.float64, device=self.device)
        return loss, {'loglik': loglik, 'kl_x': kl_x, 'entropy': z, 'kl_z_f': out[3], 'kl_z_b': z, 'info': out[7]}

    def forward(self, params, u, y, noise, condition=True, weight=1.0, local=False):
        dev = self.device
        p = {k: _f64(params[k], dev) for k in self.names}
        u, y = _f64(u, dev), _f64(y, dev)
        prob = self._problem(u.shape[0], u.shape[1], condition)
        ws = self._workspace(prob)
        with torch.no_grad():
            x0 = self._x0(p, u, y, params).contiguous()
        self._forward(p, self._constrained(p), x0, u, y, _f64(noise['eps_f'], dev), prob, ws)
        self.last_ws = ws
        red2 = None
        if self.dist is not None and not local:
            red2 = ws.out[0:2].clone()
            if weight != 1.0:
                red2.mul_(float(weight))
            all_reduce_sum(red2, self.dist)
        loss, terms = self._terms(ws, red2)
        return loss, terms, ws

    def loss_and_grads(self, params, u, y, noise, condition=True, weight=1.0, local=False):
        lib = _l.load()
        dev = self.device
        p = {k: _f64(params[k], dev) for k in self.names}
        u, y = _f64(u, dev), _f64(y, dev)
        B, T = u.shape[0], u.shape[1]
        prob = self._problem(B, T, condition)
        ws = self._workspace(prob)
        self.last_ws = ws
        c = self._constrained(p)
        eps_f = _f64(noise['eps_f'], dev)
        rp = {}
        rnames = self._recog_params(p)
        gru = None
        if self.fused_gru:
            rflat = self._recog_flat(params, p)
            gru = self._gru_forward(rflat, u, y, keep=True)
            x0 = gru['x0']
        elif rnames:
            rp = {k: p[k].detach().clone().requires_grad_(True) for k in rnames}
            x0g = self._recog(rp, u, y)
            x0 = x0g.detach().contiguous()
        else:
            x0 = self._x0(p, u, y).contiguous()
        self._forward(p, c, x0, u, y, eps_f, prob, ws)

        st = _stream()
        pb = C.byref(prob)
        cL = self.cL
        sf = self.slab_f
        red = ws.red
        lay = C.byref(self.pack_f.layout)
        N = B * self.S
        groups = (N + 15) // 16
        gB = None
        g_mode = 0
        if self.f32:
            # one launch at every tile height; the matrix section of its slab holds G = K^-1 (d loss / d K^-1) K^-1 (full up to
            # 10 row blocks, the lower block triangle of the symmetrised sum from 13: include/cbfssm_hip.h)
            g_mode = 2 if self.pack_f.layout.NBLK >= 13 else 1
            s32 = self.slab32_f
            rc = self._timed('forward_pass_adjoint', lambda: lib.cbfssm_half_forward_pass_bwd_f32(
                pb, lay, C.c_void_p(self.pack_f.buf32.data_ptr()), _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y),
                _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gx0),
                _ptr(ws.gpart_f), st))
            _l.check(rc, 'cbfssm_half_forward_pass_bwd_f32')
            if not self.stash:
                assert s32 == sf
                _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, ws.n_f, _ptr(red[:sf]), st), 'reduce f')
            else:
                # (the float64 slab of these tile heights has no matrix section: hand it to the tail as the image it expects)
                nb = self.pack_f.layout.NBLK
                nimg, og = nb * nb * 256, 2 * nb * 256
                if getattr(self, '_tmp32', None) is None:
                    self._tmp32 = torch.zeros(s32 + nimg, dtype=torch.float64, device=dev)
                t32, gB = self._tmp32[:s32], self._tmp32[s32:]
                _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), s32, ws.n_f, _ptr(t32), st), 'reduce f')
                red[:og].copy_(t32[:og])
                red[og:sf].copy_(t32[og + nimg:])
                gB.copy_(t32[og:og + nimg])
        elif not self.stash:
            rc = self._timed('forward_pass_adjoint', lambda: lib.cbfssm_half_forward_pass_bwd_f64(
                pb, lay, _ptr(self.pack_f.buf), _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y),
                _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gx0),
                _ptr(ws.gpart_f), T - 2, 0, None, None, None, 0, st))
            _l.check(rc, 'cbfssm_half_forward_pass_bwd_f64')
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, ws.n_f, _ptr(red[:sf]), st), 'reduce f')
        else:
            Mp = self.pack_f.layout.Mp
            cols_max = max(groups * 16, self.stash_bytes // (2 * Mp * 8))
            f = dict(dtype=torch.float64, device=dev)
            if self._stash_buf is None or self._stash_buf[0].numel() < Mp * cols_max:
                self._stash_buf = (torch.zeros(Mp * cols_max, **f), torch.zeros(Mp * cols_max, **f))
            sa, sk = self._stash_buf
            if getattr(self, '_contract', None) is None:
                self._contract = StashContract(self.pack_f, dev)
            self._contract.image.zero_()
            gB = self._contract.image
            tmp = torch.zeros(sf, **f)
            red[:sf].zero_()
            per = max(1, cols_max // (groups * 16))
            t_hi = T - 2
            while True:
                t_lo = max(0, t_hi - per + 1)
                cols = groups * max(0, t_hi - t_lo + 1) * 16
                rc = self._timed('forward_pass_adjoint', lambda: lib.cbfssm_half_forward_pass_bwd_f64(
                    pb, lay, _ptr(self.pack_f.buf), _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y),
                    _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gx0),
                    _ptr(ws.gpart_f), t_hi, t_lo, _ptr(ws.gx_carry), _ptr(sa), _ptr(sk), cols, st))
                _l.check(rc, 'cbfssm_half_forward_pass_bwd_f64')
                _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, groups, _ptr(tmp), st), 'reduce f')
                red[:sf] += tmp
                if cols:
                    self._timed('stash_contraction', lambda: self._contract.add(sa, sk, cols, st))
                t_hi = t_lo - 1
                if t_hi < 0:
                    break

        # data scalars and the log-likelihood's pull on var_y (cbfssmhalf.py:181-189): tail = [loglik, kl_x, 0, d/d var_y]
        tail = red[sf:]
        _l.check(lib.cbfssm_data_tail_f64(pb, _ptr(c['var_y']), _ptr(ws.ll_part), _ptr(ws.out), cL, _ptr(tail), st),
                 'cbfssm_data_tail_f64')
        gx0_b = ws.gx0.view(B, self.S, self.dim_x).sum(1)        # d loss / d x_0 per sequence (tiled over S, :87)
        rgrads = {}
        if gru is not None:
            P = gru['P']
            rc = lib.cbfssm_gru_recog_bwd_f64(B, T, self.dim_u, self.dim_y, self.dim_x, min(int(self.config['recog_len']), T), _ptr(u),
                                              _ptr(y), _ptr(rflat), _ptr(gru['act']), _ptr(gx0_b.contiguous()), _ptr(gru['gpart']), st)
            _l.check(rc, 'cbfssm_gru_recog_bwd_f64')
            rg = torch.zeros(P, dtype=torch.float64, device=dev)
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(gru['gpart']), P, B, _ptr(rg), st), 'reduce recog')
            o = 0
            for k in rnames:
                rgrads[k] = rg[o:o + p[k].numel()].view(p[k].shape)
                o += p[k].numel()
        elif rnames:
            gl = torch.autograd.grad(x0g, [rp[k] for k in rnames], grad_outputs=gx0_b)
            rgrads = dict(zip(rnames, gl))
        if self.dist is not None and not local:
            # one flat buffer per step: [slab | data scalars | stash-mode K^-1-adjoint image | recognition-model gradients]
            pieces = [red] + ([gB] if gB is not None else []) + [rgrads[k].reshape(-1) for k in rnames]
            flat = torch.cat([t.reshape(-1) for t in pieces])
            if weight != 1.0:
                flat.mul_(float(weight))
            all_reduce_sum(flat, self.dist)
            o = 0
            for t in pieces:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()

        pre = self.pre
        loss, terms = self._terms(ws, tail[0:2])
        if self.fused_tail:
            # the K_mm -> Cholesky -> K^-1 adjoint, the prior KL and the chain through the positivity transforms in HIP
            gp = [p[k].reshape(-1) for k in self.gp_names]
            pflat = getattr(params, 'flat', None)
            ngp = sum(t.numel() for t in gp)
            if pflat is None or pflat.device != dev or list(params.keys())[:7] != list(self.gp_names):
                pflat = torch.cat(gp)
            lsc = tf_forward(p[pre + 'lengthscales_unc']).reshape(-1)
            cflat = torch.cat([gp[0], gp[1], c['zvar'].reshape(-1), c['var'], lsc, c['var_x'], c['var_y']])
            if self.tail_work is None:
                nw = int(lib.cbfssm_train_tail_half_work_elems(lay))
                self.tail_work = torch.zeros(nw, dtype=torch.float64, device=dev)
            gall = torch.zeros(ngp + sum(rgrads[k].numel() for k in rnames), dtype=torch.float64, device=dev)
            rc = lib.cbfssm_train_tail_half_f64(lay, _ptr(self.pack_f.buf), _ptr(self.pack_kl.buf) if self.pack_kl is not None else None,
                                                int(lsc.numel() == 1), _ptr(red), _ptr(gB), 0, g_mode, self.dim_y, _ptr(pflat), _ptr(cflat),
                                                _ptr(self.tail_work), _ptr(gall), st)
            _l.check(rc, 'cbfssm_train_tail_half_f64')
            grads = FlatDict()
            grads.flat = gall
            o = 0
            for k in self.gp_names:
                grads[k] = gall[o:o + p[k].numel()].view(p[k].shape)
                o += p[k].numel()
            for k in rnames:
                gall[o:o + rgrads[k].numel()] = rgrads[k].reshape(-1)
                grads[k] = gall[o:o + rgrads[k].numel()].view(rgrads[k].shape)
                o += rgrads[k].numel()
            return loss, grads, terms

        grads = dict(rgrads)
        if self.f32:
            # (cross-check path) the matrix section holds G = K^-1 (d loss / d K^-1) K^-1; this restatement expects K G K
            from .train import _unpack_c
            nb, M = self.pack_f.layout.NBLK, self.M
            G = _unpack_c(gB if gB is not None else red[2 * nb * 256:2 * nb * 256 + nb * nb * 256], nb, nb)
            if g_mode == 2:
                blk = torch.arange(16 * nb, device=dev) // 16
                G = torch.where(blk[:, None] >= blk[None, :], G, torch.zeros_like(G))
            G = 0.5 * (G + G.T)[:M, :M]
            K = self.pack_f.Kmm + self.pack_f.scal[_l.SCAL_JITTER] * torch.eye(M, dtype=torch.float64, device=dev)
            Bd = torch.zeros(16 * nb, 16 * nb, dtype=torch.float64, device=dev)
            Bd[:M, :M] = K @ G @ K
            gB = Bd.view(nb, 4, 4, nb, 16).permute(0, 3, 1, 2, 4).reshape(-1)
        gz, gmu, gs2, gvar, gls, small = self._gp_adjoint(self.pack_f, red[:sf], p[pre + 'zeta_pos'], c['ls'], c['var'],
                                                          p[pre + 'zeta_mean'], c['zvar'], self.dim_x, gB, self.pack_kl)
        grads[pre + 'zeta_pos'] = gz
        grads[pre + 'zeta_mean'] = gmu
        grads[pre + 'zeta_var_unc'] = gs2 * torch.sigmoid(p[pre + 'zeta_var_unc'])
        grads[pre + 'variance_unc'] = (gvar * torch.sigmoid(p[pre + 'variance_unc'])).reshape(p[pre + 'variance_unc'].shape)
        lsu = p[pre + 'lengthscales_unc']
        if lsu.numel() == 1:
            gls = gls.sum().reshape(lsu.shape)               # shared lengthscale: its adjoint is the sum over the dims
        grads[pre + 'lengthscales_unc'] = gls * torch.sigmoid(lsu)
        grads['var_x_unc'] = small[0:self.dim_x] * torch.sigmoid(p['var_x_unc'])
        grads['var_y_unc'] = (small[16:16 + self.dim_y] + tail[3:]) * torch.sigmoid(p['var_y_unc'])
        return loss, grads, terms


class HipHalfTrainStep:
    """One `sess.run((model.train, model.loss))` of a forward-only variant (training/trainer.py:40) as ONE HIP-graph replay:
    the recognition model forward and its autograd backward (GRU(16) over recog_len steps: a few hundred tiny launches, most
    of an eager step at the small-scale shapes), K_mm / Cholesky / K^-1, the pass, its adjoint, the train tail and the Adam
    update.  One graph per (shapes, condition, GP form), captured at first use; single device (a data-parallel run keeps
    eager launches around its collective)."""

    def __init__(self, engine, opt, graph=None):
        self.engine, self.opt = engine, opt
        if graph is None:
            graph = engine.config.get('hip_graph', os.environ.get('CBFSSM_HIP_GRAPH', '1') != '0')
        self.use_graph = bool(graph) and engine.dist is None
        self._graphs = {}
        self.last_terms = self.last_ws = None

    def _eager(self, u, y, noise, condition, **kw):
        loss, grads, terms = self.engine.loss_and_grads(self.opt.views, u, y, noise, condition, **kw)
        self.opt.step(grads)
        self.last_terms, self.last_ws = terms, self.engine.last_ws
        return loss

    def step(self, u, y, noise, condition=True, **kw):
        eng = self.engine
        if not self.use_graph or kw:
            return self._eager(u, y, noise, condition, **kw)
        dev = eng.device
        u, y, eps = _f64(u, dev), _f64(y, dev), _f64(noise['eps_f'], dev)
        # (auto form: a completed condition-number read-back may flip the GP form -- other kernels, another graph)
        forms = tuple(pk.update_form() for pk in (eng.pack_f, eng.pack_kl) if pk is not None)
        key = (tuple(u.shape), tuple(y.shape), bool(condition), forms)
        g = self._graphs.get(key)
        if g is None:
            g = {'u': u.clone(), 'y': y.clone(), 'noise': {'eps_f': eps.clone()}}
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):                 # warm-up outside the capture (workspaces, autograd state); no update
                for _ in range(2):
                    eng.loss_and_grads(self.opt.views, g['u'], g['y'], g['noise'], condition)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss, grads, terms = eng.loss_and_grads(self.opt.views, g['u'], g['y'], g['noise'], condition)
                if getattr(grads, 'flat', None) is not None:
                    self.opt.step(grads)
                else:
                    self.opt.step_device(grads)
                self.opt._t -= 1                          # capture does not execute; every replay counts below
            g.update(graph=graph, loss=loss, terms=terms, ws=eng.last_ws)
            self._graphs[key] = g
        else:
            g['u'].copy_(u)
            g['y'].copy_(y)
            g['noise']['eps_f'].copy_(eps)
        g['graph'].replay()
        self.opt._t += 1
        self.last_terms, self.last_ws = g['terms'], g['ws']
        return g['loss'].clone()

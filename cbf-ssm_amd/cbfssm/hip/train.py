"""Loss + gradient of the CBF-SSM ELBO on the HIP path, and the Adam train step (cbfssm/model/cbfssm.py:273-275).

The time-loop adjoints run in the hand-written kernels (csrc/cbfssm_adjoint.hpp); this module is the glue the
north_star leaves in Python: the positivity transforms and their chain rule, the O(M^3) once-per-step adjoint of
K_mm -> Cholesky -> K^-1 (a handful of M x M float64 matmuls), the prior-KL gradient, the data-parallel
all-reduce and the optimizer update.

Multi-GPU (one process per GPU): every rank evaluates its own mini-batch shard; the ranks exchange ONE all-reduce
per step over a flat float64 buffer [reduced adjoint slab of gp_f | slab of gp_b | loglik, kl_x, entropy, ...].
The prior-KL terms and their gradients are rank-invariant and are added once after the reduce, so the result
equals a single-device evaluation of the global batch (cbfssm.py:257-261 sums over the batch).
"""
import ctypes as C
import math
import os
import torch

from . import lib as _l
from . import ops
from .ops import _ptr, _stream, _f64, tf_forward, GPPack
from .dist_utils import all_reduce_sum

PARAM_NAMES = (
    'f.zeta_pos', 'f.zeta_mean', 'f.zeta_var_unc', 'f.variance_unc', 'f.lengthscales_unc',
    'b.zeta_pos', 'b.zeta_mean', 'b.zeta_var_unc', 'b.variance_unc', 'b.lengthscales_unc',
    'var_x_unc', 'var_y_unc',
)
LOG2PI = math.log(2.0 * math.pi)


class FlatDict(dict):
    """name -> tensor views of ONE flat float64 device vector (`.flat`), in PARAM_NAMES order: lets the train-step tail
    (cbfssm_train_tail_f64, cbfssm_adam_step_f64) work on the whole parameter / gradient set in single launches."""
    flat = None


def _flat_views(flat, pl, dim_u):
    M, D, dx, dy = pl.M, pl.D, pl.dim_x, pl.dim_y
    shapes = []
    for Do in (dx, dx - dy):
        shapes += [(M, D), (M, Do), (M, Do), (1,), (D,)]
    shapes += [(dx,), (dx,)]
    out = FlatDict()
    for k, name in enumerate(PARAM_NAMES):
        n = 1
        for s in shapes[k]:
            n *= s
        out[name] = flat[pl.off[k]:pl.off[k] + n].view(*shapes[k])
    out.flat = flat
    return out


def _unpack_c(v, nrb, ncb):
    """MFMA accumulator image [rb][cb][r][lane] -> dense (16*nrb, 16*ncb); row = 16rb + (lane>>4) + 4r."""
    return v.view(nrb, ncb, 4, 4, 16).permute(0, 2, 3, 1, 4).reshape(16 * nrb, 16 * ncb)


class StashContract:
    """d loss / d K^-1 += A2bar K^T over the operand images a stash-mode launch left in HBM (cbfssm_stash_contract_f64).
    The result is an MFMA C-layout image [NBLK][NBLK][4][64]; `dense()` gives the (Mp, Mp) matrix."""

    def __init__(self, pack, device, image=None):
        self.pack = pack
        self.n = pack.layout.NBLK * pack.layout.NBLK * 256
        self.image = image if image is not None else torch.zeros(self.n, dtype=torch.float64, device=device)
        assert self.image.numel() == self.n
        self.work = None

    def add(self, sa, sk, cols, stream):
        lib = _l.load()
        nslots = cols // 16
        if nslots <= 0 or os.environ.get('CBFSSM_DIAG_SKIP_CONTRACT'):     # (diagnostic: what the contraction costs a step)
            return
        need = int(lib.cbfssm_stash_contract_work_elems(C.byref(self.pack.layout), nslots))
        if self.work is None or self.work.numel() < need:
            self.work = torch.zeros(need, dtype=torch.float64, device=self.image.device)
        _l.check(lib.cbfssm_stash_contract_f64(C.byref(self.pack.layout), _ptr(sa), _ptr(sk), nslots, _ptr(self.work),
                                               _ptr(self.image), stream), 'cbfssm_stash_contract_f64')

    def dense(self):
        nb = self.pack.layout.NBLK
        return _unpack_c(self.image, nb, nb)


class HipElboGrad:
    """loss and d loss / d (12 unconstrained tensors) for one mini-batch on one device."""

    def __init__(self, config, device, dist=None, require_adjoint=True, dtype='float64'):
        self.config = config
        self.device = torch.device(device)
        self.dist = dist
        # float32: the time loops of the forward evaluation AND of the adjoint compute in float32 (cbfssm_*_pass_f32,
        # cbfssm_*_pass_bwd_f32); K_mm / Cholesky / K^-1, their adjoint and the optimizer step stay float64, as the
        # reference's float32 models keep the Cholesky in float64 (gp_tf.py:57-65)
        assert dtype in ('float64', 'float32')
        self.f32 = dtype == 'float32'
        self.dim_u, self.dim_y, self.dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
        self.M, self.S = config['ind_pnt_num'], config['samples']
        self.D = self.dim_x + self.dim_u
        self.dob = self.dim_x - self.dim_y
        mode = ops.gp_form_mode(config) if not self.f32 else ops.gp_form_mode_f32(config)
        self.pack_f = GPPack(self.M, self.D, self.dim_x, self.device, mode)
        self.pack_b = GPPack(self.M, self.D, self.dob, self.device, mode)
        if self.f32:
            self.pack_f.cond_threshold = self.pack_b.cond_threshold = min(self.pack_f.cond_threshold, ops.F32_FORM_COND)
        self.has_adjoint = all(pk.layout.rev_slab > 0 for pk in (self.pack_f, self.pack_b))
        # tile heights above 112 inducing points run the adjoint in "stash mode" (include/cbfssm_hip.h)
        self.stash = bool(self.pack_f.layout.rev_stash)
        self.stash_bytes = int(float(config.get('adjoint_stash_gib', os.environ.get('CBFSSM_STASH_GIB', 8.0))) * 2 ** 30)
        self._stash_buf = None
        if require_adjoint:
            self._need_adjoint()
        self.slab_f = max(0, int(self.pack_f.layout.rev_slab))
        self.slab_b = max(0, int(self.pack_b.layout.rev_slab))
        if self.f32:
            # per-workgroup slabs of the float32 adjoint: the non-stash layout at every tile height (Kinvbar included)
            self.slab32_f = int(_l.load().cbfssm_rev32_slab_elems(C.byref(self.pack_f.layout)))
            self.slab32_b = int(_l.load().cbfssm_rev32_slab_elems(C.byref(self.pack_b.layout)))
        self.last_ws = None
        # flat reduce buffer: [slab_f | slab_b | loglik, kl_x, entropy, gvy_ll[dim_y] | stash mode: the two contracted
        # d loss / d K^-1 images] -- everything a data-parallel step exchanges, in ONE all-reduce
        self.ntail = 3 + self.dim_y
        self.nred = self.slab_f + self.slab_b + self.ntail
        self.nimg_f = self.pack_f.layout.NBLK ** 2 * 256 if self.stash else 0
        self.nimg_b = self.pack_b.layout.NBLK ** 2 * 256 if self.stash else 0
        self.red = torch.zeros(self.nred + self.nimg_f + self.nimg_b, dtype=torch.float64, device=self.device)
        self._ws = {}
        self.tile_pool = ops.TilePool(self.device)     # saved A2 tiles: one pool for every (B, T) this engine sees
        # train-step tail in HIP (positivity transforms, K_mm/K^-1 adjoint + prior KL, chain rule): flat vectors in
        # PARAM_NAMES order.  CBFSSM_TORCH_TAIL=1 keeps the tensor-library restatement below (same numbers, ~170 launches).
        self.pl = _l.param_layout(self.M, self.dim_x, self.dim_u, self.dim_y)
        f = dict(dtype=torch.float64, device=self.device)
        self.cflat = torch.zeros(self.pl.total, **f)
        self.gflat = torch.zeros(self.pl.total, **f)
        self.fused_tail = self.has_adjoint and not os.environ.get('CBFSSM_TORCH_TAIL')
        if self.has_adjoint:
            nw = int(_l.load().cbfssm_train_tail_work_elems(C.byref(self.pack_f.layout), C.byref(self.pack_b.layout)))
            self.tail_work = torch.zeros(max(nw, 1), **f)

    def gp_forms(self):
        """the GP forms the next launches will run, after consuming any completed condition-number read-back"""
        return tuple(pk.update_form() for pk in (self.pack_f, self.pack_b))

    def _need_adjoint(self):
        if not self.has_adjoint:
            raise _l.CbfssmHipError('no adjoint kernel for M=%d (tile height %d)' % (self.M, self.pack_f.layout.NBLK))

    # ---- forward evaluation (keeps what the adjoint needs)
    def _flat(self, params):
        """The twelve tensors as one flat vector: the optimizer's own storage when `params` are its views."""
        flat = getattr(params, 'flat', None)
        if flat is not None and flat.numel() == self.pl.total and flat.device == self.device:
            return flat
        return torch.cat([_f64(params[k], self.device).reshape(-1) for k in PARAM_NAMES])

    def _constrained_flat(self, pflat):
        """softplus + 1e-10 of every *_unc tensor in one launch; returns (p views of pflat, c dict of cflat views)."""
        _l.check(_l.load().cbfssm_constrain_f64(C.byref(self.pl), _ptr(pflat), _ptr(self.cflat), _stream()),
                 'cbfssm_constrain_f64')
        p = _flat_views(pflat, self.pl, self.dim_u)
        cv = _flat_views(self.cflat, self.pl, self.dim_u)
        c = {'var_x': cv['var_x_unc'], 'var_y': cv['var_y_unc']}
        for g in 'fb':
            c[g + '.ls'], c[g + '.var'], c[g + '.zvar'] = cv[g + '.lengthscales_unc'], cv[g + '.variance_unc'], cv[g + '.zeta_var_unc']
        return p, c

    def _constrained(self, p):
        c = {}
        for g in 'fb':
            c[g + '.ls'] = tf_forward(p[g + '.lengthscales_unc']).reshape(-1).contiguous()
            c[g + '.var'] = tf_forward(p[g + '.variance_unc']).reshape(-1).contiguous()
            c[g + '.zvar'] = tf_forward(p[g + '.zeta_var_unc']).contiguous()
        c['var_x'] = tf_forward(p['var_x_unc']).contiguous()
        c['var_y'] = tf_forward(p['var_y_unc']).contiguous()
        return c

    def forward(self, params, u, y, noise, condition=True, weight=1.0, local=False):
        """Loss only (what Trainer's test pass and Outputs fetch): returns (loss 0-d tensor, terms, workspace).
        Data parallel: `weight` scales this rank's data terms in the all-reduce (0 for a stand-in shard on a rank that
        got no sequence); `local=True` evaluates without the collective (every rank holds the whole batch)."""
        dev = self.device
        cfg = self.config
        u, y = _f64(u, dev), _f64(y, dev)
        B, T = u.shape[0], u.shape[1]
        prob = _l.make_problem(B, self.S, T, self.dim_x, self.dim_u, self.dim_y, self.M, cfg['recog_len'],
                               cfg['k_factor'], condition)
        p, c = self._constrained_flat(self._flat(params))
        ops.prepare_pair(self.pack_f, (p['f.zeta_pos'], c['f.ls'], c['f.var'], p['f.zeta_mean'], c['f.zvar']),
                         self.pack_b, (p['b.zeta_pos'], c['b.ls'], c['b.var'], p['b.zeta_mean'], c['b.zvar']))
        key = ('eval', B, T)
        if key not in self._ws:
            self._ws[key] = ops.ElboWorkspace(prob, dev, keep_h=False)
        ws = self._ws[key]
        hid_b, eps_b, eps_f = (_f64(noise[k], dev) for k in ('hid_b', 'eps_b', 'eps_f'))
        self._elbo_forward(prob, ws, c, u, y, hid_b, eps_b, eps_f)
        out = ws.out
        if self.dist is not None and not local:
            # data terms summed over the ranks' shards, prior KL counted once (cbfssm.py:257-261)
            red = out[0:3].clone()
            if weight != 1.0:
                red.mul_(float(weight))
            all_reduce_sum(red, self.dist)
            lf = cfg['loss_factors']
            cL, cE = float(lf[0]) / self.S, float(lf[1]) / self.S
            loss = -(red[0] * cL - red[1] * cL + red[2] * cE - out[3] - out[4])
            terms = {'loglik': red[0], 'kl_x': red[1], 'entropy': red[2], 'kl_z_f': out[3], 'kl_z_b': out[4],
                     'info': out[7]}
        else:
            loss = out[6].clone()       # (the terms below stay views into the workspace: the next evaluation overwrites them)
            terms = {'loglik': out[0], 'kl_x': out[1], 'entropy': out[2], 'kl_z_f': out[3], 'kl_z_b': out[4],
                     'info': out[7]}
        self.last_ws = ws
        return loss, terms, ws

    def _workspace(self, prob):
        key = (prob.B, prob.T)
        if key not in self._ws:
            # (the float32 adjoint recomputes the kernel tile and A2: no saved tiles)
            ws = ops.ElboWorkspace(prob, self.device, keep_h=True, packs=None if self.f32 else (self.pack_f, self.pack_b),
                                   pool=self.tile_pool)
            lib = _l.load()
            n_f = int(lib.cbfssm_rev_workgroups(C.byref(prob), 0))
            n_b = int(lib.cbfssm_rev_workgroups(C.byref(prob), 1))
            f = dict(dtype=torch.float64, device=self.device)
            ws.gy2 = torch.zeros_like(ws.y2)
            if self.f32:
                # every step's [A2 | kernel tile] registers of the float32 passes, float32: the adjoint reads them back instead
                # of recomputing both (cbfssm_rev32.hip, KSV); same pool and budget as the float64 tiles, None above it
                if not os.environ.get('CBFSSM_F32_NO_TILES'):
                    na_f = (int(lib.cbfssm_saved_a2_f32_elems(C.byref(prob), C.byref(self.pack_f.layout), 0)) + 1) // 2
                    na_b = (int(lib.cbfssm_saved_a2_f32_elems(C.byref(prob), C.byref(self.pack_b.layout), 1)) + 1) // 2
                    assert na_f >= 0 and na_b > 0, 'record count'
                    reserve = 8.0 * prob.T * prob.B * prob.S * (3 * prob.dim_x + 8 * self.dob) + 2.0 * 2 ** 30
                    ws.a2s_f, ws.a2s_b = self.tile_pool.get(max(na_f, 1), na_b, reserve)
                ws.gpart_f = torch.zeros((n_f + 32) * self.slab32_f, **f)
                ws.gpart_b = torch.zeros((n_b + 32) * self.slab32_b, **f)
                ws.n_f, ws.n_b = n_f, n_b
                self._ws[key] = ws
                return ws
            if self.stash:
                n_b = 2 * n_f          # one launch per segment range, grid.z = 1
                ws.gx_carry = torch.zeros(prob.B * prob.S, prob.dim_x, **f)
            ws.gpart_f = torch.zeros((n_f + 32) * self.slab_f, **f)      # + CBFSSM_REDUCE_SPLIT scratch slabs
            ws.gpart_b = torch.zeros((n_b + 32) * self.slab_b, **f)
            ws.n_f, ws.n_b = n_f, n_b
            self._ws[key] = ws
        return self._ws[key]

    def loss_and_grads(self, params, u, y, noise, condition=True, weight=1.0, local=False):
        """params: dict of unconstrained float64 device tensors.  Returns (loss 0-d tensor, grads dict, terms).
        `weight`, `local`: as in forward()."""
        s = self._grads_local(params, u, y, noise, condition)
        if not local:
            self._grads_collective(s, weight)
        return self._grads_finish(s)

    # A train step in three parts, so that a data-parallel step can replay the device work on either side of the
    # collective from HIP graphs (HipTrainStep): everything up to this rank's reduced sums `red`, the all-reduce, the
    # once-per-step tail.
    def _grads_collective(self, s, weight=1.0):
        if self.dist is not None:
            if weight != 1.0:
                self.red.mul_(float(weight))
            all_reduce_sum(self.red, self.dist)     # the ONE collective of a train step (RCCL over xGMI); in stash mode
                                                    # the contracted K^-1-adjoint images ride in the same buffer

    def _grads_local(self, params, u, y, noise, condition=True):
        self._need_adjoint()
        lib = _l.load()
        dev = self.device
        cfg = self.config
        u, y = _f64(u, dev), _f64(y, dev)
        B, T = u.shape[0], u.shape[1]
        prob = _l.make_problem(B, self.S, T, self.dim_x, self.dim_u, self.dim_y, self.M, cfg['recog_len'],
                               cfg['k_factor'], condition)
        pflat = self._flat(params)
        p, c = self._constrained_flat(pflat)
        ops.prepare_pair(self.pack_f, (p['f.zeta_pos'], c['f.ls'], c['f.var'], p['f.zeta_mean'], c['f.zvar']),
                         self.pack_b, (p['b.zeta_pos'], c['b.ls'], c['b.var'], p['b.zeta_mean'], c['b.zvar']))
        ws = self._workspace(prob)
        self.last_ws = ws
        hid_b, eps_b, eps_f = (_f64(noise[k], dev) for k in ('hid_b', 'eps_b', 'eps_f'))
        lf = cfg['loss_factors']
        self._elbo_forward(prob, ws, c, u, y, hid_b, eps_b, eps_f)

        # ---- adjoint time loops
        st = _stream()
        pb = C.byref(prob)
        cL, cE = float(lf[0]) / self.S, float(lf[1]) / self.S
        red = self.red
        sf, sb = self.slab_f, self.slab_b
        gB_f = gB_b = None
        split = self._split(prob) if not self.f32 else self._split(prob, adjoint=False)
        if self.f32:
            gB_f, gB_b = self._adjoint_f32(prob, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE, red, split)
        elif not self.stash and split is not None:
            self._adjoint_split(prob, split, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE)
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, ws.n_f, _ptr(red[:sf]), st), 'reduce f')
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_b), sb, ws.n_b, _ptr(red[sf:sf + sb]), st), 'reduce b')
        elif not self.stash:
            rc = lib.cbfssm_forward_pass_bwd_f64(pb, C.byref(self.pack_f.layout), _ptr(self.pack_f.buf),
                                                 _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y), _ptr(ws.y2),
                                                 _ptr(eps_f) if eps_f.numel() else None, _ptr(ws.x),
                                                 _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gy2),
                                                 _ptr(ws.gpart_f), st)
            _l.check(rc, 'cbfssm_forward_pass_bwd_f64')
            rc = lib.cbfssm_backward_pass_bwd_f64(pb, C.byref(self.pack_b.layout), _ptr(self.pack_b.buf),
                                                  _ptr(c['var_x']), _ptr(u), _ptr(y), _ptr(hid_b), _ptr(eps_b),
                                                  _ptr(ws.h_all), _ptr(ws.fmv_b), _ptr(ws.a2s_b), _ptr(ws.gy2), cE,
                                                  _ptr(ws.gpart_b), st)
            _l.check(rc, 'cbfssm_backward_pass_bwd_f64')
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, ws.n_f, _ptr(red[:sf]), st), 'reduce f')
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_b), sb, ws.n_b, _ptr(red[sf:sf + sb]), st), 'reduce b')
        else:
            gB_f, gB_b = self._adjoint_stash(prob, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE, red)
        # ---- data scalars and the log-likelihood's pull on var_y (cbfssm.py:245-251)
        tail = red[sf + sb:sf + sb + self.ntail]
        _l.check(lib.cbfssm_data_tail_f64(pb, _ptr(c['var_y']), _ptr(ws.ll_part), _ptr(ws.out), cL, _ptr(tail), st),
                 'cbfssm_data_tail_f64')
        return dict(ws=ws, p=p, c=c, pflat=pflat, gB_f=gB_f, gB_b=gB_b, cL=cL, cE=cE)

    def _grads_finish(self, s):
        lib = _l.load()
        dev = self.device
        st = _stream()
        ws, p, c, pflat, gB_f, gB_b, cL, cE = (s[k] for k in ('ws', 'p', 'c', 'pflat', 'gB_f', 'gB_b', 'cL', 'cE'))
        red = self.red
        sf, sb = self.slab_f, self.slab_b
        tail = red[sf + sb:sf + sb + self.ntail]
        loglik, kl_x, entropy = tail[0], tail[1], tail[2]
        kl_z_f, kl_z_b = self.pack_f.scal[_l.SCAL_KLZ], self.pack_b.scal[_l.SCAL_KLZ]
        if self.dist is not None:
            loss = -(loglik * cL - kl_x * cL + entropy * cE - kl_z_f - kl_z_b)             # cbfssm.py:258-261
        else:
            loss = ws.out[6].clone()                    # the same combination, done by cbfssm_elbo_combine_f64
        terms = {'loglik': loglik, 'kl_x': kl_x, 'entropy': entropy, 'kl_z_f': kl_z_f, 'kl_z_b': kl_z_b,
                 'info': ws.out[7]}
        if self.fused_tail:
            Mp = self.pack_f.layout.Mp
            # float32 adjoint: the matrix section is G = K^-1 (d loss / d K^-1) K^-1 (full up to 10 row blocks, the
            # lower-triangular blocks of G + G^T above: include/cbfssm_hip.h)
            g_mode = 0 if not self.f32 else (2 if self.pack_f.layout.NBLK >= 13 else 1)
            rc = lib.cbfssm_train_tail_g_f64(C.byref(self.pl), C.byref(self.pack_f.layout), _ptr(self.pack_f.buf),
                                             C.byref(self.pack_b.layout), _ptr(self.pack_b.buf), _ptr(red), _ptr(gB_f),
                                             _ptr(gB_b), 0, g_mode, _ptr(pflat), _ptr(self.cflat), _ptr(self.tail_work),
                                             _ptr(self.gflat), st)
            _l.check(rc, 'cbfssm_train_tail_g_f64')
            return loss, _flat_views(self.gflat, self.pl, self.dim_u), terms

        # ---- once-per-step adjoints and the chain through the positivity transforms (tensor-library restatement)
        grads = {}
        gvx = torch.zeros(self.dim_x, dtype=torch.float64, device=dev)
        gvy = torch.zeros(self.dim_x, dtype=torch.float64, device=dev)
        if self.f32:
            # the float32 adjoint's matrix section holds G = K^-1 (d loss / d K^-1) K^-1 (include/cbfssm_hip.h); this
            # restatement expects d loss / d K^-1 = K G K, K = K_mm + jitter I
            def kinv_adjoint(img, pack):
                nb, M = pack.layout.NBLK, pack.M
                G = _unpack_c(img, nb, nb)
                if nb >= 13:             # lower-triangular blocks of G + G^T (diagonal blocks: of G)
                    blk = torch.arange(16 * nb, device=dev) // 16
                    G = torch.where(blk[:, None] >= blk[None, :], G, torch.zeros_like(G))
                G = 0.5 * (G + G.T)[:M, :M]
                K = pack.Kmm + pack.scal[_l.SCAL_JITTER] * torch.eye(M, dtype=torch.float64, device=dev)
                Bd = torch.zeros(16 * nb, 16 * nb, dtype=torch.float64, device=dev)
                Bd[:M, :M] = K @ G @ K
                return Bd.view(nb, 4, 4, nb, 16).permute(0, 3, 1, 2, 4).reshape(-1)
            nbf, nbb = self.pack_f.layout.NBLK, self.pack_b.layout.NBLK
            gB_f = kinv_adjoint(gB_f if gB_f is not None else red[2 * nbf * 256:2 * nbf * 256 + nbf * nbf * 256], self.pack_f)
            gB_b = kinv_adjoint(gB_b if gB_b is not None else red[sf + 2 * nbb * 256:sf + 2 * nbb * 256 + nbb * nbb * 256],
                                self.pack_b)
        for g, pack, slab, Do, gBx in (('f', self.pack_f, red[:sf], self.dim_x, gB_f),
                                       ('b', self.pack_b, red[sf:sf + sb], self.dob, gB_b)):
            gz, gmu, gs2, gvar, gls, small = self._gp_adjoint(pack, slab, p[g + '.zeta_pos'], c[g + '.ls'], c[g + '.var'],
                                                              p[g + '.zeta_mean'], c[g + '.zvar'], Do, gBx)
            grads[g + '.zeta_pos'] = gz
            grads[g + '.zeta_mean'] = gmu
            grads[g + '.zeta_var_unc'] = gs2 * torch.sigmoid(p[g + '.zeta_var_unc'])
            grads[g + '.variance_unc'] = (gvar * torch.sigmoid(p[g + '.variance_unc'])).reshape(p[g + '.variance_unc'].shape)
            grads[g + '.lengthscales_unc'] = gls * torch.sigmoid(p[g + '.lengthscales_unc'])
            gvx[:Do] += small[0:Do]
            if g == 'f':
                gvy += small[16:16 + self.dim_x]
        gvy[:self.dim_y] += tail[3:]
        grads['var_x_unc'] = gvx * torch.sigmoid(p['var_x_unc'])
        grads['var_y_unc'] = gvy * torch.sigmoid(p['var_y_unc'])

        return loss, grads, terms

    # ---- chain-group split: chains never interact, so a pass can be issued in two pieces on two HIP streams.  When the
    # number of 16-chain groups is not a multiple of the CU count (C3: 320 groups on 256 CUs), the one-workgroup-per-
    # group forward-direction kernels would leave 3/4 of the chip idle in their last round; the remainder groups are
    # run early and the many-workgroup backward-run kernels of the other piece fill the idle CUs meanwhile.
    def _split(self, prob, adjoint=True):
        # (the stash-mode adjoint has its own launch schedule; its forward evaluation still splits)
        if (self.stash and adjoint) or os.environ.get('CBFSSM_NO_SPLIT'):
            return None
        groups = (prob.B * prob.S + 15) // 16
        force = os.environ.get('CBFSSM_SPLIT_MAIN')            # tests: force a split at this group index
        if force:
            return (int(force), groups - int(force)) if 0 < int(force) < groups else None
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        if groups <= ncu or groups % ncu == 0:
            return None
        main = (groups // ncu) * ncu
        return main, groups - main

    def _sub_problem(self, prob, g0, ng):
        q = _l.Problem()
        C.memmove(C.byref(q), C.byref(prob), C.sizeof(_l.Problem))
        q.group0, q.ngroups = int(g0), int(ng)
        return q

    def _side_stream(self):
        if getattr(self, '_s1', None) is None:
            self._s1 = torch.cuda.Stream(device=self.device)
        return self._s1

    def _elbo_forward(self, prob, ws, c, u, y, hid_b, eps_b, eps_f):
        lf = self.config['loss_factors']
        split = self._split(prob, adjoint=False)
        if split is None:
            ops.elbo_forward(prob, self.pack_f, self.pack_b, c['var_x'], c['var_y'], u, y, hid_b, eps_b, eps_f, lf, ws,
                             f32=self.f32)
            return
        lib = _l.load()
        main, rest = split
        p_main, p_rest = self._sub_problem(prob, 0, main), self._sub_problem(prob, main, rest)
        s0, s1 = torch.cuda.current_stream(), self._side_stream()
        st0, st1 = C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)
        lb, lf_ = C.byref(self.pack_b.layout), C.byref(self.pack_f.layout)
        e_eps = _ptr(eps_f) if eps_f.numel() else None

        if self.f32:
            b32b, b32f = (C.c_void_p(pk.pack_f32().data_ptr()) for pk in (self.pack_b, self.pack_f))   # (re-packed on s0)

        def bwd(q, st):
            if self.f32:
                _l.check(lib.cbfssm_backward_pass_f32(C.byref(q), lb, b32b, _ptr(c['var_x']), _ptr(u), _ptr(y), _ptr(hid_b),
                                                      _ptr(eps_b), _ptr(ws.y2), _ptr(ws.h_all), _ptr(ws.fmv_b),
                                                      _ptr(ws.a2s_b), _ptr(ws.ent_part), st), 'cbfssm_backward_pass_f32')
                return
            _l.check(lib.cbfssm_backward_pass_f64(C.byref(q), lb, _ptr(self.pack_b.buf), _ptr(c['var_x']), _ptr(u), _ptr(y),
                                                  _ptr(hid_b), _ptr(eps_b), _ptr(ws.y2), _ptr(ws.h_all),
                                                  _ptr(ws.fmv_b), _ptr(ws.a2s_b), _ptr(ws.ent_part), st),
                     'cbfssm_backward_pass_f64')

        def fwd(q, st):
            if self.f32:
                _l.check(lib.cbfssm_forward_pass_f32(C.byref(q), lf_, b32f, _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y),
                                                     _ptr(ws.y2), e_eps, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f),
                                                     _ptr(ws.kl_part), st),
                         'cbfssm_forward_pass_f32')
                return
            _l.check(lib.cbfssm_forward_pass_f64(C.byref(q), lf_, _ptr(self.pack_f.buf), _ptr(c['var_x']), _ptr(c['var_y']),
                                                 _ptr(u), _ptr(y), _ptr(ws.y2), e_eps, _ptr(ws.x), _ptr(ws.fmv_f),
                                                 _ptr(ws.a2s_f), _ptr(ws.kl_part), st), 'cbfssm_forward_pass_f64')
        e0 = torch.cuda.Event()
        e0.record(s0)
        bwd(p_rest, st0)                       # the remainder groups first ...
        fwd(p_rest, st0)                       # ... their (few, long) forward workgroups overlap
        s1.wait_event(e0)
        bwd(p_main, st1)                       # ... the many short backward-run workgroups of the main piece
        e1 = torch.cuda.Event()
        e1.record(s1)
        s0.wait_event(e1)
        fwd(p_main, st0)
        pb = C.byref(prob)
        rc = lib.cbfssm_loglik_moments_f64(pb, _ptr(c['var_y']), _ptr(y), _ptr(ws.x), _ptr(ws.ll_part),
                                           _ptr(ws.pred_mean), _ptr(ws.pred_var), _ptr(ws.int_mean), _ptr(ws.int_var), st0)
        _l.check(rc, 'cbfssm_loglik_moments_f64')
        rc = lib.cbfssm_elbo_combine_f64(pb, float(lf[0]), float(lf[1]), _ptr(ws.ll_part), ws.ll_part.numel(),
                                         _ptr(ws.kl_part), ws.kl_part.numel(), _ptr(ws.ent_part), ws.ent_part.numel(),
                                         _ptr(self.pack_f.scal), _ptr(self.pack_b.scal), _ptr(ws.out), st0)
        _l.check(rc, 'cbfssm_elbo_combine_f64')

    def _adjoint_split(self, prob, split, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE):
        lib = _l.load()
        main, rest = split
        p_main, p_rest = self._sub_problem(prob, 0, main), self._sub_problem(prob, main, rest)
        s0, s1 = torch.cuda.current_stream(), self._side_stream()
        st0, st1 = C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)
        lb, lf_ = C.byref(self.pack_b.layout), C.byref(self.pack_f.layout)
        e_eps = _ptr(eps_f) if eps_f.numel() else None

        def rfwd(q, st):
            _l.check(lib.cbfssm_forward_pass_bwd_f64(C.byref(q), lf_, _ptr(self.pack_f.buf), _ptr(c['var_x']),
                                                     _ptr(c['var_y']), _ptr(u), _ptr(y), _ptr(ws.y2), e_eps, _ptr(ws.x),
                                                     _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gy2), _ptr(ws.gpart_f), st), 'cbfssm_forward_pass_bwd_f64')

        def rbwd(q, st):
            _l.check(lib.cbfssm_backward_pass_bwd_f64(C.byref(q), lb, _ptr(self.pack_b.buf), _ptr(c['var_x']), _ptr(u),
                                                      _ptr(y), _ptr(hid_b), _ptr(eps_b), _ptr(ws.h_all), _ptr(ws.fmv_b),
                                                      _ptr(ws.a2s_b), _ptr(ws.gy2), cE, _ptr(ws.gpart_b), st),
                     'cbfssm_backward_pass_bwd_f64')
        rfwd(p_main, st0)                      # a whole number of rounds over the CUs
        e2 = torch.cuda.Event()
        e2.record(s0)
        rfwd(p_rest, st0)                      # few long workgroups ...
        s1.wait_event(e2)
        rbwd(p_main, st1)                      # ... next to the main piece's backward-run adjoint
        e3 = torch.cuda.Event()
        e3.record(s1)
        rbwd(p_rest, st0)
        s0.wait_event(e3)

    def _adjoint_f32(self, prob, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE, red, split=None):
        """float32 adjoint (cbfssm_*_pass_bwd_f32): one call per direction (above 208 inducing points a call is two passes
        over the time loop), float64 slabs in the non-stash layout reduced by the float64 reduction.  For the tile heights
        whose float64 adjoint runs in stash mode the Kinvbar section of the reduced slab is handed to the train tail the
        way the stash contraction's images are."""
        lib = _l.load()
        pb = C.byref(prob)
        st = _stream()
        b32f, b32b = self.pack_f.buf32, self.pack_b.buf32           # filled by the forward evaluation (ops.elbo_forward)
        sf, sb = self.slab_f, self.slab_b
        e_eps = _ptr(eps_f) if eps_f.numel() else None

        def rfwd(q, stq):
            _l.check(lib.cbfssm_forward_pass_bwd_f32(C.byref(q), C.byref(self.pack_f.layout), C.c_void_p(b32f.data_ptr()),
                                                     _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y), _ptr(ws.y2), e_eps,
                                                     _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gy2), _ptr(ws.gpart_f), stq),
                     'cbfssm_forward_pass_bwd_f32')

        def rbwd(q, stq):
            _l.check(lib.cbfssm_backward_pass_bwd_f32(C.byref(q), C.byref(self.pack_b.layout), C.c_void_p(b32b.data_ptr()),
                                                      _ptr(c['var_x']), _ptr(u), _ptr(y), _ptr(hid_b), _ptr(eps_b),
                                                      _ptr(ws.h_all), _ptr(ws.fmv_b), _ptr(ws.a2s_b), _ptr(ws.gy2), cE, _ptr(ws.gpart_b), stq),
                     'cbfssm_backward_pass_bwd_f32')
        if split is None:
            rfwd(prob, st)
            rbwd(prob, st)
        else:
            # the chain-group split of the float64 adjoint (_adjoint_split): a main piece of whole rounds over the CUs, the
            # remainder's forward-pass adjoint next to the main piece's many-workgroup backward-run adjoint
            main, rest = split
            p_main, p_rest = self._sub_problem(prob, 0, main), self._sub_problem(prob, main, rest)
            s0, s1 = torch.cuda.current_stream(), self._side_stream()
            st0, st1 = C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)
            rfwd(p_main, st0)
            e2 = torch.cuda.Event()
            e2.record(s0)
            rfwd(p_rest, st0)
            s1.wait_event(e2)
            rbwd(p_main, st1)
            e3 = torch.cuda.Event()
            e3.record(s1)
            rbwd(p_rest, st0)
            s0.wait_event(e3)
        if not self.stash:
            assert self.slab32_f == sf and self.slab32_b == sb
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, ws.n_f, _ptr(red[:sf]), st), 'reduce f')
            _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_b), sb, ws.n_b, _ptr(red[sf:sf + sb]), st), 'reduce b')
            # (the matrix section holds G = K^-1 (d loss / d K^-1) K^-1, the data part of d loss / d K_mm itself:
            #  cbfssm_train_tail_g_f64 takes it as it is, _grads_finish passes g_mode)
            return None, None
        if getattr(self, '_tmp32', None) is None:
            self._tmp32 = torch.zeros(self.slab32_f + self.slab32_b, dtype=torch.float64, device=self.device)
        tf, tb = self._tmp32[:self.slab32_f], self._tmp32[self.slab32_f:]
        _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), self.slab32_f, ws.n_f, _ptr(tf), st), 'reduce f')
        _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_b), self.slab32_b, ws.n_b, _ptr(tb), st), 'reduce b')
        o = self.nred
        imgs = []
        for t32, lo, n, nimg, io in ((tf, 0, sf, self.nimg_f, o), (tb, sf, sb, self.nimg_b, o + self.nimg_f)):
            nb = int(round((nimg // 256) ** 0.5))
            og = 2 * nb * 256                                        # [mubar | s2bar] precede the Kinvbar images
            red[lo:lo + og].copy_(t32[:og])
            red[lo + og:lo + n].copy_(t32[og + nimg:])
            img = red[io:io + nimg]
            img.copy_(t32[og:og + nimg])
            imgs.append(img)
        return imgs[0], imgs[1]

    def _adjoint_stash(self, prob, ws, c, u, y, hid_b, eps_b, eps_f, cL, cE, red):
        """Stash-mode adjoint (M > 112): time-chunked launches, every launch followed by one float64 GEMM that
        contracts the stashed A2bar / K tiles into d loss / d K^-1 (include/cbfssm_hip.h).

        Two streams: the forward-pass adjoint walks t downwards in chunks on the current stream; a backward-run launch
        (a range of resample-to-resample segments of both runs) needs the y2 adjoint only for its own time range, so it
        starts on the side stream as soon as the forward-pass adjoint has passed below that range and fills the CUs the
        one-workgroup-per-chain-group forward-pass adjoint leaves idle (320 workgroups on 256 CUs at C4)."""
        lib = _l.load()
        pb = C.byref(prob)
        dev = self.device
        f = dict(dtype=torch.float64, device=dev)
        Mp = self.pack_f.layout.Mp
        T, N = prob.T, prob.B * prob.S
        groups = (N + 15) // 16
        R = prob.recog_len
        P = 2 * R
        # half of the stash budget per direction (the two directions are in flight together)
        cols_f = max(groups * 16, self.stash_bytes // (4 * Mp * 8))
        cols_b = max(groups * 16 * 2 * P, self.stash_bytes // (4 * Mp * 8))
        if self._stash_buf is None or self._stash_buf[0].numel() < Mp * cols_f or self._stash_buf[2].numel() < Mp * cols_b:
            self._stash_buf = (torch.zeros(Mp * cols_f, **f), torch.zeros(Mp * cols_f, **f),
                               torch.zeros(Mp * cols_b, **f), torch.zeros(Mp * cols_b, **f))
        sa_f, sk_f, sa_b, sk_b = self._stash_buf
        sf, sb = self.slab_f, self.slab_b
        nseg = int(lib.cbfssm_bwd_segments(pb))
        per_b = max(1, cols_b // (groups * 2 * P * 16))
        # measurement hook (bench.py): with a list in self._prof every launch of this schedule is bracketed by HIP events on
        # ONE stream (no overlap between the two directions), so that each kernel's own time can be summed per kind
        prof = getattr(self, '_prof', None)
        overlap = not os.environ.get('CBFSSM_NO_SPLIT') and prof is None
        s0 = torch.cuda.current_stream()
        s1 = self._side_stream() if overlap else s0
        st0, st1 = C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)
        red[:sf + sb].zero_()
        tmp_f, tmp_b = torch.zeros(max(sf, 1), **f), torch.zeros(max(sb, 1), **f)
        if getattr(self, '_contract', None) is None:
            o = self.nred
            self._contract = (StashContract(self.pack_f, dev, red[o:o + self.nimg_f]),
                              StashContract(self.pack_b, dev, red[o + self.nimg_f:o + self.nimg_f + self.nimg_b]))
        con_f, con_b = self._contract
        con_f.image.zero_()
        con_b.image.zero_()
        e_eps = _ptr(eps_f) if eps_f.numel() else None
        if overlap:
            s1.wait_stream(s0)                               # the forward evaluation and the zeroing above

        def timed(kind, stream, fn):
            if prof is None:
                return fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            prof.append((kind, e0, e1))

        def rfwd(t_hi, t_lo):
            """forward-pass adjoint of steps t_hi .. t_lo (descending) in launches that fit the stash, on s0"""
            per = max(1, cols_f // (groups * 16))
            while True:
                lo = max(t_lo, t_hi - per + 1)
                cols = groups * max(0, t_hi - lo + 1) * 16
                timed('forward_pass_adjoint', s0, lambda: _l.check(lib.cbfssm_forward_pass_bwd_ex_f64(
                    pb, C.byref(self.pack_f.layout), _ptr(self.pack_f.buf), _ptr(c['var_x']), _ptr(c['var_y']), _ptr(u), _ptr(y),
                    _ptr(ws.y2), e_eps, _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), cL, _ptr(ws.gy2), _ptr(ws.gpart_f), t_hi, lo,
                    _ptr(ws.gx_carry), _ptr(sa_f), _ptr(sk_f), cols, st0), 'cbfssm_forward_pass_bwd_ex_f64'))
                ev = torch.cuda.Event()
                ev.record(s0)                                # gy2[t] is final for every t > lo (and t = 0 once lo = 0)

                def red_f():
                    _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_f), sf, groups, _ptr(tmp_f[:sf]), st0), 'reduce f')
                    red[:sf] += tmp_f[:sf]
                timed('reductions', s0, red_f)
                if cols:
                    timed('stash_contraction', s0, lambda: con_f.add(sa_f, sk_f, cols, st0))
                if lo <= t_lo or t_hi < 0:
                    return ev
                t_hi = lo - 1

        def rbwd(seg0, seg1):
            """both backward runs, segments [seg0, seg1), on s1"""
            cols = groups * 2 * (seg1 - seg0) * P * 16
            with torch.cuda.stream(s1):
                timed('backward_pass_adjoint', s1, lambda: _l.check(lib.cbfssm_backward_pass_bwd_ex_f64(
                    pb, C.byref(self.pack_b.layout), _ptr(self.pack_b.buf), _ptr(c['var_x']), _ptr(u), _ptr(y), _ptr(hid_b),
                    _ptr(eps_b), _ptr(ws.h_all), _ptr(ws.fmv_b), _ptr(ws.a2s_b), _ptr(ws.gy2), cE, _ptr(ws.gpart_b), seg0, seg1, 1,
                    _ptr(sa_b), _ptr(sk_b), cols, st1), 'cbfssm_backward_pass_bwd_ex_f64'))

                def red_b():
                    _l.check(lib.cbfssm_reduce_partials_f64(_ptr(ws.gpart_b), sb, 2 * groups, _ptr(tmp_b[:sb]), st1), 'reduce b')
                    red[sf:sf + sb] += tmp_b[:sb]
                timed('reductions', s1, red_b)
                timed('stash_contraction', s1, lambda: con_b.add(sa_b, sk_b, cols, st1))

        t_hi = T - 2                                         # next forward-pass-adjoint step to process
        seg1 = nseg
        done_all = (t_hi < 0)
        if done_all:
            rfwd(-1, 0)                                      # T == 1: the launch that only hands out the x_0 adjoints
        while seg1 > 0:
            seg0 = max(0, seg1 - per_b)
            # run 1 reaches furthest down: its segment seg0 starts at max(0, P seg0 - R); the y2 adjoint of time t is
            # written by forward-pass-adjoint step t - 1 (t = 0: by step 0)
            tmin = 0 if seg0 <= 0 else max(0, P * seg0 - R)
            need_lo = max(0, tmin - 1)
            if not done_all and t_hi >= need_lo:
                ev = rfwd(t_hi, need_lo)
                t_hi = need_lo - 1
                done_all = (t_hi < 0)
                if overlap:
                    s1.wait_event(ev)
            rbwd(seg0, seg1)
            seg1 = seg0
        if not done_all:
            rfwd(t_hi, 0)
        if overlap:
            s0.wait_stream(s1)
        return con_f.image, con_b.image

    def _gp_adjoint(self, pack, slab, Z, ls, var, zmean, zvar, Do, gB_stash=None, kl_pack=None):
        """Adjoint of gp_prepare (K_mm -> chol -> K^-1, operand scaling, prior KL) given the reduced data slab."""
        lay = pack.layout
        M, D, NBLK, JB = self.M, self.D, lay.NBLK, lay.JB
        o = 0
        gmu = _unpack_c(slab[o:o + NBLK * 256], NBLK, 1)[:M, :Do]
        o += NBLK * 256
        gs2 = _unpack_c(slab[o:o + NBLK * 256], NBLK, 1)[:M, :Do]
        o += NBLK * 256
        if gB_stash is None:
            gB = _unpack_c(slab[o:o + NBLK * NBLK * 256], NBLK, NBLK)[:M, :M]
        else:
            gB = _unpack_c(gB_stash, NBLK, NBLK)[:M, :M]           # C-layout image of cbfssm_stash_contract_f64
        if not self.stash:
            o += NBLK * NBLK * 256                                 # (the slab of a non-stash tile height has the section)
        gZf = _unpack_c(slab[o:o + NBLK * JB * 256], NBLK, JB)
        o += NBLK * JB * 256
        small = slab[o:o + 128]
        Kinv, Kmm = pack.Kinv, pack.Kmm
        Zs = pack.section('Zs', (M, D))
        gZt = gZf[:M, :D] - Zs * gZf[:M, D:D + 1]           # Ebar x~^T - z~ o rowsum(Ebar)
        glx = small[32:32 + D]
        gsig, glogsig = small[96], small[97]
        # prior KL (gp_tf.py:163-172), weight 1 in the loss, added once (rank invariant).  PR-SSM factorises the prior
        # without jitter (prssm.py:81-82): its KL terms then use the jitter-free inverse of kl_pack.
        Kkl = kl_pack.Kinv if kl_pack is not None else Kinv
        gmu = gmu + Kkl @ zmean
        gs2 = gs2 + 0.5 * (torch.diagonal(Kkl)[:, None] - 1.0 / zvar)
        gB_kl = 0.5 * (torch.diag(zvar.sum(1)) + zmean @ zmean.T)
        # K^-1 = (K_mm + jitter I)^-1 ; log det term of the KL: d/dK (0.5 Do log det K) = 0.5 Do K^-1
        jit = pack.scal[_l.SCAL_JITTER]
        if kl_pack is None:
            T = Kinv @ (gB + gB_kl)
            TK = T @ Kinv
            gK = -TK + 0.5 * Do * Kinv
            # tr(Kbar K_mm) with K^-1 K_mm = I - jitter K^-1: the terms are of order cond |G|, where the entry sum of
            # Kbar o K_mm cancels numbers of order cond^2 |G| (it decides d loss / d sigma^2 on an ill-conditioned K_mm)
            tr_gKK = -torch.trace(T) + jit * torch.trace(TK) + 0.5 * Do * (M - jit * torch.trace(Kinv))
        else:
            T = Kinv @ gB
            TK = T @ Kinv
            gK = -TK - (Kkl @ gB_kl @ Kkl) + 0.5 * Do * Kkl
            # (the prior's K_kl^-1 is jitter free: K_kl^-1 K_mm = I)
            tr_gKK = -torch.trace(T) + jit * torch.trace(TK) - (Kkl * gB_kl.T).sum() + 0.5 * Do * M
        # K = var * exp(-0.5 d2(z~))                                           (gp_tf.py:33-49)
        gKK = gK * Kmm
        gvar = tr_gKK / var[0] + gsig + glogsig / var[0]
        Wd = -0.5 * gKK
        Ws = Wd + Wd.T
        gZt = gZt + 2.0 * (Ws.sum(1)[:, None] * Zs - Ws @ Zs)
        gz = gZt / ls
        gls = -((gZt * Zs).sum(0) + glx) / ls
        return gz, gmu, gs2, gvar, gls, small


class TFAdam:
    """tf.train.AdamOptimizer's update rule (TF 1.8): lr_t = lr sqrt(1-b2^t)/(1-b1^t); p -= lr_t m/(sqrt(v)+eps)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.names = list(params.keys())
        sizes = [params[k].numel() for k in self.names]
        dev = params[self.names[0]].device
        self.flat = torch.cat([params[k].reshape(-1) for k in self.names]).to(torch.float64).contiguous()
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.gflat = torch.zeros_like(self.flat)
        self.views, self.gviews = FlatDict(), {}
        self.views.flat = self.flat
        o = 0
        for k, n in zip(self.names, sizes):
            self.views[k] = self.flat[o:o + n].view(params[k].shape)
            self.gviews[k] = self.gflat[o:o + n].view(params[k].shape)
            o += n
        self.lr, self.b1, self.b2, self.eps = float(lr), beta1, beta2, eps
        self.t_dev = torch.zeros(1, dtype=torch.float64, device=dev)      # step counter as the update kernel sees it
        self._t = 0

    @property
    def t(self):
        return self._t

    @t.setter
    def t(self, value):
        self._t = int(value)
        self.t_dev.fill_(float(value))

    def step(self, grads):
        gflat = getattr(grads, 'flat', None)
        if gflat is not None and gflat.is_cuda and gflat.numel() == self.flat.numel():
            # the whole update in one launch on the flat vectors; the step counter lives on the device, so the call is
            # the same inside a captured HIP graph
            rc = _l.load().cbfssm_adam_step_f64(self.flat.numel(), _ptr(self.flat), _ptr(gflat), _ptr(self.m),
                                                _ptr(self.v), _ptr(self.t_dev), self.lr, self.b1, self.b2, self.eps,
                                                _stream())
            _l.check(rc, 'cbfssm_adam_step_f64')
            self._t += 1
            return
        for k in self.names:
            self.gviews[k].copy_(grads[k])
        self.t = self._t + 1
        g = self.gflat
        self.m.mul_(self.b1).add_(g, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        self.flat.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-lr_t)

    def step_device(self, grads):
        """The same update with the step counter and lr_t on the device: no host value enters, so the whole step can be
        captured in a HIP graph and replayed."""
        for k in self.names:
            self.gviews[k].copy_(grads[k])
        self.t_dev.add_(1.0)
        self._t += 1
        g = self.gflat
        self.m.mul_(self.b1).add_(g, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        b1t = torch.pow(torch.full_like(self.t_dev, self.b1), self.t_dev)
        b2t = torch.pow(torch.full_like(self.t_dev, self.b2), self.t_dev)
        lr_t = self.lr * torch.sqrt(1.0 - b2t) / (1.0 - b1t)
        self.flat.sub_(lr_t * self.m / (self.v.sqrt() + self.eps))

    def state_dict(self):
        return {'flat': self.flat.clone(), 'm': self.m.clone(), 'v': self.v.clone(), 't': self.t, 'names': self.names}

    def load_state_dict(self, sd):
        self.flat.copy_(sd['flat'])
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.t = int(sd['t'])


class HipTrainStep:
    """One `sess.run((model.train, model.loss))` (training/trainer.py:40): loss, gradient, Adam update."""

    def __init__(self, config, params, device, dist=None, graph=None, engine=None, opt=None, dtype='float64'):
        """Either builds its own engine and optimiser from (config, params), or drives the pair a model object already
        owns (`engine`, `opt`: cbfssm.model.CBFSSM)."""
        if engine is None:
            engine = HipElboGrad(config, device, dist, dtype=dtype)
            opt = TFAdam({k: _f64(params[k], device).clone() for k in PARAM_NAMES}, config['learning_rate'])
        self.engine, self.opt = engine, opt
        self.params = self.opt.views
        # HIP graph: a train step is ~60 launches (kernels of this library + the small torch ops of the once-per-step
        # adjoints and Adam); the small workloads (C1, C2) are launch-bound without it.  One graph per (shapes,
        # condition), captured at first use.  Not in stash mode (its launch schedule depends on the stash budget).
        if graph is None:
            graph = config.get('hip_graph', os.environ.get('CBFSSM_HIP_GRAPH', '1') != '0')
        # With a process group the collective stays eager between two graphs (local part / tail + Adam).
        self.use_graph = bool(graph) and not self.engine.stash
        self._graphs = {}
        self.last_ws = None

    def step(self, u, y, noise, condition=True, weight=1.0, local=False):
        if self.use_graph and not local:
            return self._graph_step(u, y, noise, condition, weight)
        loss, grads, terms = self.engine.loss_and_grads(self.params, u, y, noise, condition, weight=weight, local=local)
        self.opt.step(grads)
        self.last_terms = terms
        self.last_ws = self.engine.last_ws
        return loss

    def _graph_step(self, u, y, noise, condition, weight=1.0):
        dev = self.engine.device
        u, y = _f64(u, dev), _f64(y, dev)
        # (auto form: a completed condition-number read-back may flip the GP form -- a different set of kernels, so a
        #  different graph)
        key = (tuple(u.shape), tuple(y.shape), bool(condition), self.engine.gp_forms())
        g = self._graphs.get(key)
        names = ('hid_b', 'eps_b', 'eps_f')
        if g is None:
            g = {'u': u.clone(), 'y': y.clone(), 'noise': {k: _f64(noise[k], dev).clone() for k in names}}
            # warm-up outside the capture (workspaces, kernel attributes); the parameters are not touched
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                # (without the collective: the ranks do not capture in the same step -- a rank sees a new shard shape
                #  when the others replay -- so a warm-up all-reduce would be one collective too many on this rank)
                self.engine.loss_and_grads(self.params, g['u'], g['y'], g['noise'], condition, local=True)
            cur.wait_stream(side)
            eng = self.engine
            front = None
            t_before = self.opt._t
            # with a process group, other threads of this process (the collective library's completion polling) may
            # touch the runtime while this thread captures: only this thread's calls are checked then
            mode = 'global' if eng.dist is None else 'thread_local'
            try:
                if eng.dist is not None:
                    # nothing of the warm-up's collective may still be in flight (its completion polling runs on another
                    # thread of this process) while the streams are in capture mode
                    torch.cuda.synchronize(dev)
                    front = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(front, capture_error_mode=mode):
                        state = eng._grads_local(self.params, g['u'], g['y'], g['noise'], condition)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode=mode):
                    if front is None:
                        state = eng._grads_local(self.params, g['u'], g['y'], g['noise'], condition)
                    loss, grads, terms = eng._grads_finish(state)
                    if getattr(grads, 'flat', None) is not None:
                        self.opt.step(grads)
                    else:
                        self.opt.step_device(grads)
                    self.opt._t -= 1              # capture does not execute; every replay counts below
            except RuntimeError as e:
                if eng.dist is None:
                    raise
                # a data-parallel rank that cannot capture runs the same launches eagerly: the collective sequence is
                # the same in both modes, so the other ranks are not affected
                import warnings
                warnings.warn('HIP graph capture failed on this rank (%s): continuing with eager launches' % e)
                torch.cuda.synchronize(dev)
                self.opt._t = t_before
                self.use_graph = False
                return self.step(u, y, noise, condition, weight=weight)
            g.update(graph=graph, front=front, state=state, loss=loss, terms=terms, ws=state['ws'])
            self._graphs[key] = g
        else:
            g['u'].copy_(u)
            g['y'].copy_(y)
            for k in names:
                g['noise'][k].copy_(_f64(noise[k], dev))
        if g['front'] is not None:
            g['front'].replay()
            self.engine._grads_collective(g['state'], weight)
        g['graph'].replay()
        self.opt._t += 1
        self.last_terms = g['terms']
        self.last_ws = g['ws']            # the workspace THIS graph writes (a later eval pass may have moved engine.last_ws)
        return g['loss'].clone()


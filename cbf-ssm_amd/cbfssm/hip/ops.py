"""Tensor-level wrappers around the C ABI (PyTorch is used for device memory and streams only).

All tensors are float64, contiguous, on one CUDA(=HIP) device.  Trajectories are time-major on the device
(x: (T,N,dim_x), y2: (T,N,dim_x-dim_y), N = B*S, chain c = b*S + s); `as_btsd` gives the reference's (B,T,S,d) view.
"""
import ctypes as C
import os
import torch

from . import lib as _l


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), 'need contiguous float64 CUDA tensor'
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f64(t, device):
    return torch.as_tensor(t, dtype=torch.float64, device=device).contiguous()


F32_FORM_COND = 1e3


def gp_form_mode_f32(config=None, bf16=False):
    """GP form policy of a float32 engine: the float64 one ('dense' | 'tri' | 'auto', default auto) with the automatic
    switch to the reference's two triangular products (gp_tf.py:137-145) at cond(K_mm + jitter I) > 1e3 instead of 3e7
    (`F32_FORM_COND`; GPPack.cond_threshold): in float32 the dense form's fvar_0 = sigma^2 - k.(K^-1 k) cancels to
    cond eps_32, the sum of squares sigma^2 - |L^-1 k|^2 does not.  Measured at the C5 shape against float64 (pred_var):
    cond 9e5: dense 1.6e-2, two-triangular 9.5e-3; cond 5e7: 0.21 / 0.24 -- past cond ~1e5 float32 itself (fmean carries
    cond eps_32 in either form and the recurrence amplifies it over 1000 steps) is the limit, not the form.  The bf16-operand
    probe rounds the operands of the dense contraction and stays dense."""
    if bf16:
        return 'dense'
    return gp_form_mode(config)


def gp_form_mode(config=None):
    """'dense' | 'tri' | 'auto' (config['gp_form'] or CBFSSM_GP_FORM; default auto): which form of GPModel.predict the pass
    kernels run.  dense: A2 = K^-1 k in one product, fvar_0 = sigma^2 - k.A2.  tri: the reference's own order
    (gp_tf.py:137-145), A = L^-1 k, fvar_0 = sigma^2 - |A|^2, A2 = L^-T A as two triangular products.  auto: dense while
    the measured infinity-norm condition number of K_mm + jitter I (about 3.4 x the 2-norm one on the trained-like
    family) stays below CBFSSM_GP_FORM_COND (3e7), tri above.  Where that number comes from: the dense form's
    fvar_0 = sigma^2 - k.(K^-1 k) loses about 8e-14 x cond_2 of (fvar + var_x) near the inducing inputs (measured: 5.9e-9 at
    1.1e5, 2.5e-7 at 2.4e6, 3.2e-6 at 4e7, 6.0e-6 at 2e8), the north_star tolerance is 1e-5: the switch
    at cond_2 ~ 1e7 keeps the dense form ten times inside it, and through the full C3 recurrence the dense form is then
    at 1.4e-8 on the predictive variance (cond 2e6).  Round 2 switched at 1e5, which ran every trained-like model 6-24 %
    slower for digits nobody asked for."""
    mode = None
    if config is not None:
        mode = config.get('gp_form')
    mode = mode or os.environ.get('CBFSSM_GP_FORM', 'auto')
    if mode not in ('dense', 'tri', 'auto'):
        raise ValueError("gp_form must be 'dense', 'tri' or 'auto', not %r" % (mode,))
    return mode


class GPPack:
    """Loop-invariant operands of one GPModel on the device (include/cbfssm_hip.h: cbfssm_gp_prepare_f64)."""

    def __init__(self, M, D, Do, device, form_mode=None):
        self.layout = _l.pack_layout(M, D, Do)
        self.M, self.D, self.Do = M, D, Do
        self.buf = torch.zeros(self.layout.total, dtype=torch.float64, device=device)
        self.form_mode = form_mode or gp_form_mode()
        self.cond_threshold = float(os.environ.get('CBFSSM_GP_FORM_COND', 3e7))
        self.layout.gp_form = _l.GP_FORM_TRI if self.form_mode == 'tri' else _l.GP_FORM_DENSE
        self._cond_host = None          # pinned landing slots of the asynchronous condition-number read-backs
        self._decided = self.form_mode != 'auto'
        self.last_cond = None

    # ---- form policy (auto mode).  The condition number comes out of the prepare kernel on the device; reading it is an
    # asynchronous copy into pinned memory.  The lag is DETERMINISTIC: the call of step k starts the read-back of step k's
    # prepare and consumes the one started at step k - LAG (waiting on that event: two steps old, it has completed long
    # ago, so no step waits for the host) -- two identically seeded runs switch kernels in the same step, and so do the
    # ranks of a data-parallel run (same parameters => same condition numbers => same decision, no extra collective).
    # Only the very first decision blocks (there is nothing older to go by).
    LAG = 2

    def gp_form(self):
        return 'tri' if self.layout.gp_form == _l.GP_FORM_TRI else 'dense'

    def _consume_cond(self, cond):
        self.last_cond = cond
        thr = self.cond_threshold
        tri = self.layout.gp_form == _l.GP_FORM_TRI
        # (hysteresis: a factor four between switching up and switching back, every switch is another captured graph)
        if not (cond == cond) or cond > thr:
            tri = True
        elif cond < 0.25 * thr:
            tri = False
        self.layout.gp_form = _l.GP_FORM_TRI if tri else _l.GP_FORM_DENSE
        self._decided = True

    def update_form(self, blocking=False):
        """Start the read-back of the latest prepare's condition number and consume the one that is LAG calls old (it may
        flip the form); returns the form the next launches run.  `blocking`: wait for the read-back of the latest prepare."""
        if self.form_mode != 'auto' or torch.cuda.is_current_stream_capturing():
            return self.gp_form()
        if self._cond_host is None:
            self._cond_host = torch.zeros(self.LAG + 2, dtype=torch.float64).pin_memory()
            self._cond_q = []           # [(slot, event)] oldest first
            self._cond_slot = 0
        # (in the launch stream: a side stream for this copy was measured -- the extra event traffic costs more than the
        #  copy's 15 us: C1 train step 0.600 -> 0.620 ms, C3 10.88 -> 10.90)
        slot = self._cond_slot
        self._cond_slot = (slot + 1) % self._cond_host.numel()
        self._cond_host[slot:slot + 1].copy_(self.scal[_l.SCAL_COND:_l.SCAL_COND + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._cond_q.append((slot, ev))
        if blocking or not self._decided:
            ev.synchronize()
            self._cond_q = []
            self._consume_cond(float(self._cond_host[slot]))
        elif len(self._cond_q) > self.LAG:
            oslot, oev = self._cond_q.pop(0)
            oev.synchronize()
            self._consume_cond(float(self._cond_host[oslot]))
        return self.gp_form()

    def prepare(self, Z, lengthscales, variance, zeta_mean, zeta_var, jitter=_l.JITTER):
        dev = self.buf.device
        Z, ls, var = _f64(Z, dev), _f64(lengthscales, dev).reshape(-1), _f64(variance, dev).reshape(-1)
        zm, zv = _f64(zeta_mean, dev), _f64(zeta_var, dev)
        assert Z.shape == (self.M, self.D) and ls.numel() == self.D and var.numel() == 1
        assert zm.shape == (self.M, self.Do) and zv.shape == (self.M, self.Do)
        rc = _l.load().cbfssm_gp_prepare_f64(C.byref(self.layout), _ptr(Z), _ptr(ls), _ptr(var), _ptr(zm), _ptr(zv),
                                             float(jitter), _ptr(self.buf), _stream())
        _l.check(rc, 'cbfssm_gp_prepare_f64')
        self.update_form()                      # (blocks for the very first decision only)
        return self

    def section(self, name, shape):
        off = getattr(self.layout, name)
        n = 1
        for s in shape:
            n *= s
        return self.buf[off:off + n].view(*shape)

    @property
    def scal(self):
        return self.section('scal', (_l.SCAL_COUNT,))

    @property
    def L(self):
        return self.section('L', (self.M, self.M))

    @property
    def Kmm(self):
        return self.section('Kmm', (self.M, self.M))

    @property
    def Kinv(self):
        return self.section('Kinv', (self.M, self.M))

    def pack_f32(self, bf16=False):
        """float32 MFMA images of this pack for the float32-arithmetic passes (cbfssm_gp_pack_f32): the Cholesky and K^-1
        were computed in float64 and are cast here, as the reference does for float32 models (gp_tf.py:57-65).
        bf16=True: the K^-1 operand rounded to bfloat16 (cbfssm_gp_pack_bf16; the passes then round the kernel tile too)."""
        lib = _l.load()
        if getattr(self, 'buf32', None) is None:
            n = int(lib.cbfssm_pack_f32_elems(C.byref(self.layout)))
            self.buf32 = torch.zeros(n, dtype=torch.float32, device=self.buf.device)
        fn = lib.cbfssm_gp_pack_bf16 if bf16 else lib.cbfssm_gp_pack_f32
        _l.check(fn(C.byref(self.layout), _ptr(self.buf), C.c_void_p(self.buf32.data_ptr()), _stream()), 'cbfssm_gp_pack_f32')
        return self.buf32

    def predict_f32(self, X):
        X = _f64(X, self.buf.device)
        assert X.dim() == 2 and X.shape[1] == self.D
        n = X.shape[0]
        fmean = torch.empty(n, self.Do, dtype=torch.float64, device=X.device)
        fvar = torch.empty_like(fmean)
        b32 = self.pack_f32()
        rc = _l.load().cbfssm_gp_predict_f32(C.byref(self.layout), C.c_void_p(b32.data_ptr()), _ptr(X), n, _ptr(fmean),
                                             _ptr(fvar), _stream())
        _l.check(rc, 'cbfssm_gp_predict_f32')
        return fmean, fvar

    def predict(self, X):
        X = _f64(X, self.buf.device)
        assert X.dim() == 2 and X.shape[1] == self.D
        n = X.shape[0]
        fmean = torch.empty(n, self.Do, dtype=torch.float64, device=X.device)
        fvar = torch.empty_like(fmean)
        rc = _l.load().cbfssm_gp_predict_f64(C.byref(self.layout), _ptr(self.buf), _ptr(X), n, _ptr(fmean),
                                             _ptr(fvar), _stream())
        _l.check(rc, 'cbfssm_gp_predict_f64')
        return fmean, fvar


def prepare_pair(pack0, args0, pack1, args1, jitter=_l.JITTER):
    """gp_f and gp_b in ONE launch (one workgroup each): args = (Z, lengthscales, variance, zeta_mean, zeta_var)."""
    ptrs = []
    keep = []
    for pack, args in ((pack0, args0), (pack1, args1)):
        dev = pack.buf.device
        Z, ls, var, zm, zv = args
        Z, ls, var = _f64(Z, dev), _f64(ls, dev).reshape(-1), _f64(var, dev).reshape(-1)
        zm, zv = _f64(zm, dev), _f64(zv, dev)
        assert Z.shape == (pack.M, pack.D) and ls.numel() == pack.D and var.numel() == 1
        assert zm.shape == (pack.M, pack.Do) and zv.shape == (pack.M, pack.Do)
        keep.append((Z, ls, var, zm, zv))
        ptrs += [C.byref(pack.layout), _ptr(Z), _ptr(ls), _ptr(var), _ptr(zm), _ptr(zv), _ptr(pack.buf)]
    rc = _l.load().cbfssm_gp_prepare2_f64(*ptrs, float(jitter), _stream())
    _l.check(rc, 'cbfssm_gp_prepare2_f64')
    for pack in (pack0, pack1):
        pack.update_form()


def kmm_chol(Z, lengthscales, variance, jitter=_l.JITTER):
    """RBF.K(zeta_pos) and cast_cholesky (gp_tf.py:33-65,129-130) -> (Kmm, L, info)."""
    dev = Z.device
    Z = _f64(Z, dev)
    M, D = Z.shape
    ls, var = _f64(lengthscales, dev).reshape(-1), _f64(variance, dev).reshape(-1)
    Kmm = torch.empty(M, M, dtype=torch.float64, device=dev)
    L = torch.empty_like(Kmm)
    info = torch.zeros(1, dtype=torch.float64, device=dev)
    work = torch.empty(2 * M * M + M * D, dtype=torch.float64, device=dev)
    rc = _l.load().cbfssm_kmm_chol_f64(M, D, _ptr(Z), _ptr(ls), _ptr(var), float(jitter), _ptr(Kmm), _ptr(L),
                                       _ptr(info), _ptr(work), _stream())
    _l.check(rc, 'cbfssm_kmm_chol_f64')
    return Kmm, L, info


def as_btsd(x_tnd, B, S):
    """(T, N, d) time-major device layout -> the reference's (B, T, S, d) view (cbfssm.py:95,181)."""
    T, N, d = x_tnd.shape
    return x_tnd.view(T, B, S, d).permute(1, 0, 2, 3)


def a2s_budget_bytes(device):
    """HBM the saved A2 tiles of an engine may take: CBFSSM_A2S_MAX_GB, by default 3/4 of what is free right now."""
    cap = os.environ.get('CBFSSM_A2S_MAX_GB')
    if cap is not None:
        return float(cap) * 2 ** 30
    return 0.75 * torch.cuda.mem_get_info(torch.device(device))[0]


class TilePool:
    """The saved A2 tiles of an engine: ONE pair of flat buffers, sized for the largest (B, T) seen, that every
    workspace views into (a partial last mini-batch, a test pass at another batch size and a retrain at another
    seq_len all reuse the same HBM; only one evaluation is in flight per engine).  The budget is taken once, when the
    first buffers are allocated.  A later shape that needs MORE than the pool holds gets new, larger buffers while the
    budget allows (the old ones stay alive: captured HIP graphs hold their addresses) and is logged; a shape that does
    not fit the budget runs its adjoint in recompute mode, and says so."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs = None            # (a2s_f, a2s_b or None)
        self.retired = []
        self.budget = None
        self.log = []

    def get(self, n_f, n_b, reserve=0.0):
        """views of n_f / n_b doubles (n_b = 0: no backward GP), or (None, None) when the tiles do not fit the budget.
        `reserve`: bytes the caller still has to allocate after this (the rest of the workspace): they must fit next to
        the tiles, or the very next allocation would fail instead of the adjoint falling back to recompute mode."""
        import warnings
        if self.bufs is not None and self.bufs[0].numel() >= n_f and (n_b == 0 or (self.bufs[1] is not None and
                                                                                  self.bufs[1].numel() >= n_b)):
            return self.bufs[0][:max(n_f, 1)], (self.bufs[1][:max(n_b, 1)] if n_b else None)
        if self.budget is None:
            self.budget = a2s_budget_bytes(self.device)
        have = 0 if self.bufs is None else 8.0 * (self.bufs[0].numel() + (self.bufs[1].numel() if self.bufs[1] is not None else 0))
        need = 8.0 * (n_f + n_b)
        free_now = torch.cuda.mem_get_info(self.device)[0]
        # the budget covers everything the pool holds: the buffers a growth step retires stay allocated (captured graphs
        # hold their addresses) and count; what is free must also hold the caller's remaining buffers
        held = float(self.bytes())
        if held + need > self.budget or need + reserve > 0.9 * free_now:
            msg = ('saved A2 tiles of this shape need %.2f GB, over the budget of %.2f GB (CBFSSM_A2S_MAX_GB): its adjoint '
                   'recomputes A2 (slower, same numbers)' % (need / 2 ** 30, self.budget / 2 ** 30))
            self.log.append(msg)
            warnings.warn(msg)
            return None, None
        if self.bufs is not None:
            msg = ('saved-A2 pool grows from %.2f to %.2f GB for a larger shape; the old buffers stay allocated for the '
                   'graphs captured on them' % (have / 2 ** 30, need / 2 ** 30))
            self.log.append(msg)
            warnings.warn(msg)
            self.retired.append(self.bufs)
        f = dict(dtype=torch.float64, device=self.device)
        self.bufs = (torch.zeros(max(n_f, 1), **f), torch.zeros(max(n_b, 1), **f) if n_b else None)
        return self.bufs[0][:max(n_f, 1)], (self.bufs[1][:max(n_b, 1)] if n_b else None)

    def bytes(self):
        tot = 0
        for pair in ([self.bufs] if self.bufs is not None else []) + self.retired:
            tot += 8 * (pair[0].numel() + (pair[1].numel() if pair[1] is not None else 0))
        return tot


class ElboWorkspace:
    """Device buffers of one ELBO evaluation for fixed (B, T) -- allocated once, reused every step."""

    def __init__(self, prob, device, keep_h=False, packs=None, pool=None):
        p = prob
        N, T = p.B * p.S, p.T
        dob = p.dim_x - p.dim_y
        f = dict(dtype=torch.float64, device=device)
        lib = _l.load()
        self.n_ent = int(lib.cbfssm_backward_pass_partials(C.byref(p)))
        self.n_kl = int(lib.cbfssm_forward_pass_partials(C.byref(p)))
        self.y2 = torch.zeros(T, N, dob, **f)
        self.h_all = torch.zeros(2, T, N, dob, **f) if keep_h else None
        # every step's (fmean, fvar): the adjoint reads them instead of recomputing the predictive products
        self.fmv_b = torch.zeros(2, T, N, dob, 2, **f) if keep_h else None
        self.fmv_f = torch.zeros(max(T - 1, 0), N, p.dim_x, 2, **f) if keep_h else None
        # every step's A2 = K^-1 k tiles: with them the adjoint skips one of its three M x M x 16 products per step.
        # Kept while both buffers fit the budget: CBFSSM_A2S_MAX_GB, by default three quarters of the HBM that is free
        # right now -- the part has 288 GB and this is what it is for (C5: 196 GB of tiles, train step 1770 -> 1604 ms);
        # otherwise the adjoint recomputes A2.
        self.a2s_f = self.a2s_b = None
        if keep_h and packs is not None:
            n_f = int(lib.cbfssm_saved_a2_elems(C.byref(p), C.byref(packs[0].layout), 0))
            n_b = int(lib.cbfssm_saved_a2_elems(C.byref(p), C.byref(packs[1].layout), 1)) if packs[1] is not None else 0
            if pool is None:
                pool = TilePool(device)
            # what this workspace allocates after the tiles: trajectories, (fmean, fvar), adjoint buffers, slabs / stash
            reserve = 8.0 * T * N * (3 * p.dim_x + 8 * dob) + 2.0 * 2 ** 30
            self.a2s_f, self.a2s_b = pool.get(n_f, n_b, reserve)     # views into the engine's one pool (or None, None)
        self.x = torch.zeros(T, N, p.dim_x, **f)
        self.ent_part = torch.zeros(self.n_ent, **f)
        self.kl_part = torch.zeros(self.n_kl, **f)
        self.ll_part = torch.zeros(int(lib.cbfssm_loglik_partials(C.byref(p))), **f)   # [block][dim_y]
        self.pred_mean = torch.zeros(p.B, T, p.dim_y, **f)
        self.pred_var = torch.zeros(p.B, T, p.dim_y, **f)
        self.int_mean = torch.zeros(p.B, T, p.dim_x, **f)
        self.int_var = torch.zeros(p.B, T, p.dim_x, **f)
        self.out = torch.zeros(8, **f)


def elbo_forward(prob, pack_f, pack_b, var_x, var_y, u, y, hid_b, eps_b, eps_f, loss_factors, ws=None,
                 keep_h=False, f32=False, bf16=False):
    """One forward evaluation of the ELBO (cbfssm.py:84-271) from prepared GP packs.  Asynchronous.

    Returns the workspace; ws.out = [loglik, kl_x, entropy, kl_z_f, kl_z_b, elbo, loss, info].
    """
    lib = _l.load()
    dev = u.device
    if ws is None:
        ws = ElboWorkspace(prob, dev, keep_h)
    st = _stream()
    pb = C.byref(prob)
    N = prob.B * prob.S
    assert u.shape == (prob.B, prob.T, prob.dim_u) and y.shape == (prob.B, prob.T, prob.dim_y)
    assert hid_b.numel() == 2 * prob.T * N and eps_b.numel() == 2 * prob.T * N
    assert eps_f.numel() == (prob.T - 1) * N
    if f32:
        # float32 arithmetic in the time loops (cbfssm_*_pass_f32), float64 storage; forward evaluation only
        b32, f32p = pack_b.pack_f32(bf16), pack_f.pack_f32(bf16)
        rc = lib.cbfssm_backward_pass_f32(pb, C.byref(pack_b.layout), C.c_void_p(b32.data_ptr()), _ptr(var_x), _ptr(u),
                                          _ptr(y), _ptr(hid_b), _ptr(eps_b), _ptr(ws.y2), _ptr(ws.h_all), _ptr(ws.fmv_b),
                                          _ptr(ws.a2s_b), _ptr(ws.ent_part), st)
        _l.check(rc, 'cbfssm_backward_pass_f32')
        rc = lib.cbfssm_forward_pass_f32(pb, C.byref(pack_f.layout), C.c_void_p(f32p.data_ptr()), _ptr(var_x), _ptr(var_y),
                                         _ptr(u), _ptr(y), _ptr(ws.y2), _ptr(eps_f) if eps_f.numel() else None,
                                         _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), _ptr(ws.kl_part), st)
        _l.check(rc, 'cbfssm_forward_pass_f32')
        return _elbo_tail(lib, pb, prob, pack_f, pack_b, var_y, y, loss_factors, ws, st)
    rc = lib.cbfssm_backward_pass_f64(pb, C.byref(pack_b.layout), _ptr(pack_b.buf), _ptr(var_x), _ptr(u), _ptr(y),
                                      _ptr(hid_b), _ptr(eps_b), _ptr(ws.y2), _ptr(ws.h_all), _ptr(ws.fmv_b),
                                      _ptr(ws.a2s_b), _ptr(ws.ent_part), st)
    _l.check(rc, 'cbfssm_backward_pass_f64')
    rc = lib.cbfssm_forward_pass_f64(pb, C.byref(pack_f.layout), _ptr(pack_f.buf), _ptr(var_x), _ptr(var_y),
                                     _ptr(u), _ptr(y), _ptr(ws.y2), _ptr(eps_f) if eps_f.numel() else None,
                                     _ptr(ws.x), _ptr(ws.fmv_f), _ptr(ws.a2s_f), _ptr(ws.kl_part), st)
    _l.check(rc, 'cbfssm_forward_pass_f64')
    return _elbo_tail(lib, pb, prob, pack_f, pack_b, var_y, y, loss_factors, ws, st)


def _elbo_tail(lib, pb, prob, pack_f, pack_b, var_y, y, loss_factors, ws, st):
    rc = lib.cbfssm_loglik_moments_f64(pb, _ptr(var_y), _ptr(y), _ptr(ws.x), _ptr(ws.ll_part), _ptr(ws.pred_mean),
                                       _ptr(ws.pred_var), _ptr(ws.int_mean), _ptr(ws.int_var), st)
    _l.check(rc, 'cbfssm_loglik_moments_f64')
    rc = lib.cbfssm_elbo_combine_f64(pb, float(loss_factors[0]), float(loss_factors[1]),
                                     _ptr(ws.ll_part), ws.ll_part.numel(), _ptr(ws.kl_part), ws.kl_part.numel(),
                                     _ptr(ws.ent_part), ws.ent_part.numel(), _ptr(pack_f.scal), _ptr(pack_b.scal),
                                     _ptr(ws.out), st)
    _l.check(rc, 'cbfssm_elbo_combine_f64')
    return ws


def tf_forward(x):
    """softplus(x) + 1e-10, the positivity transform of every constrained quantity (tf_transform.py:19-21)."""
    return torch.nn.functional.softplus(x, beta=1.0, threshold=1e9) + 1e-10


class HipElbo:
    """Forward-only ELBO evaluator on one device from the twelve unconstrained tensors (no autograd)."""

    def __init__(self, config, device, dtype='float64'):
        """dtype 'float32': the time loops compute in float32 (cbfssm_*_pass_f32; the reference's model dtype argument,
        cbfssm.py:12, with the Cholesky kept in float64, gp_tf.py:57-65); storage and the ELBO reductions stay float64."""
        self.config = config
        self.device = torch.device(device)
        assert dtype in ('float64', 'float32', 'bfloat16')
        self.f32 = dtype in ('float32', 'bfloat16')
        self.bf16 = dtype == 'bfloat16'       # bf16 OPERANDS of the K^-1 K contraction inside the float32 passes
        self.dim_u, self.dim_y, self.dim_x = config['ds'].dim_u, config['ds'].dim_y, config['dim_x']
        self.M, self.S = config['ind_pnt_num'], config['samples']
        D = self.dim_x + self.dim_u
        mode = gp_form_mode(config) if not self.f32 else gp_form_mode_f32(config, self.bf16)
        self.pack_f = GPPack(self.M, D, self.dim_x, self.device, mode)
        self.pack_b = GPPack(self.M, D, self.dim_x - self.dim_y, self.device, mode)
        if self.f32:
            self.pack_f.cond_threshold = self.pack_b.cond_threshold = min(self.pack_f.cond_threshold, F32_FORM_COND)
        self._ws = {}

    def gp_form(self):
        return self.pack_f.gp_form() + '/' + self.pack_b.gp_form()

    def prepare(self, params):
        p = {k: _f64(v, self.device) for k, v in params.items()}
        args = [(p[g + '.zeta_pos'], tf_forward(p[g + '.lengthscales_unc']), tf_forward(p[g + '.variance_unc']),
                 p[g + '.zeta_mean'], tf_forward(p[g + '.zeta_var_unc'])) for g in 'fb']
        prepare_pair(self.pack_f, args[0], self.pack_b, args[1])
        for pk in (self.pack_f, self.pack_b):
            pk.update_form(blocking=True)       # an explicit prepare(): the evaluation that follows uses THIS K_mm's form
        self.var_x = tf_forward(p['var_x_unc']).contiguous()
        self.var_y = tf_forward(p['var_y_unc']).contiguous()

    def problem(self, B, T, condition):
        c = self.config
        return _l.make_problem(B, self.S, T, self.dim_x, self.dim_u, self.dim_y, self.M, c['recog_len'],
                               c['k_factor'], condition)

    def run(self, u, y, noise, condition=True, keep_h=False):
        u, y = _f64(u, self.device), _f64(y, self.device)
        B, T = u.shape[0], u.shape[1]
        prob = self.problem(B, T, condition)
        key = (B, T, keep_h)
        if key not in self._ws:
            self._ws[key] = ElboWorkspace(prob, self.device, keep_h)
        ws = self._ws[key]
        hid_b, eps_b, eps_f = (_f64(noise[k], self.device) for k in ('hid_b', 'eps_b', 'eps_f'))
        elbo_forward(prob, self.pack_f, self.pack_b, self.var_x, self.var_y, u, y, hid_b, eps_b, eps_f,
                     self.config['loss_factors'], ws, f32=self.f32, bf16=self.bf16)
        return ws


class NoisePipeline:
    """Standard-normal draws for the next ELBO evaluation, generated on a side stream while the current evaluation
    runs (the noise does not depend on the parameters): two buffers, one being consumed, one being filled.
    One normal per (b, s) and step, as the reference tiles it (cbfssm.py:134,149,209)."""

    def __init__(self, device, generator=None, with_backward=True):
        """The draws come from the library's own generator (cbfssm_normal_f64: Philox4x32-10 + Box-Muller, keyed by the
        generator's seed, one running element offset per pipeline: a run is reproducible from its seed, whatever the
        order of buffer sizes); CBFSSM_TORCH_NOISE=1 draws with torch's generator instead."""
        self.device = torch.device(device)
        self.gen = generator
        self.with_backward = with_backward
        self.stream = torch.cuda.Stream(device=self.device)
        self._bufs = {}
        self._ready = {}       # key -> (index of the buffer that holds / is receiving fresh noise, event)
        self.use_lib = not os.environ.get('CBFSSM_TORCH_NOISE')
        self.seed = int(generator.initial_seed()) & (2 ** 64 - 1) if generator is not None else int.from_bytes(os.urandom(8), 'little')
        self.offset = 0        # elements drawn so far

    def _numel(self, T, N):
        return (4 * T * N if self.with_backward else 0) + (T - 1) * N

    def _fill(self, key, idx):
        buf = self._bufs[key][idx]
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)          # the buffer's previous consumer (issued on `cur`) must be done
        with torch.cuda.stream(self.stream):
            if self.use_lib:
                rc = _l.load().cbfssm_normal_f64(self.seed, self.offset, buf.numel(), _ptr(buf),
                                                 C.c_void_p(self.stream.cuda_stream))
                _l.check(rc, 'cbfssm_normal_f64')
                self.offset += buf.numel()
            else:
                buf.normal_(generator=self.gen)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._ready[key] = (idx, ev)

    def next(self, T, N):
        key = (T, N)
        if key not in self._bufs:
            n = self._numel(T, N)
            self._bufs[key] = [torch.empty(n, dtype=torch.float64, device=self.device) for _ in range(2)]
            self._fill(key, 0)
        idx, ev = self._ready[key]
        torch.cuda.current_stream(self.device).wait_event(ev)
        buf = self._bufs[key][idx]
        self._fill(key, 1 - idx)               # start the draw for the next call
        if self.with_backward:
            a = 2 * T * N
            return {'hid_b': buf[:a], 'eps_b': buf[a:2 * a], 'eps_f': buf[2 * a:]}
        return {'eps_f': buf}

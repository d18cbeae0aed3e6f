"""Diagnostic: the float32 train step with the kept [A2 | kernel tile] records against the recomputing path at a shape whose
backward-run record buffer exceeds 2^31 floats (M = 300: 10 240 floats per step and group).
usage: python profiles/tools/f32_tiles_check.py [T] [B]"""
import os
import sys
import dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'cbf-ssm_amd')):
    sys.path.insert(0, p)
import torch
from cbfssm import synthetic as syn
from cbfssm.hip.train import HipElboGrad, PARAM_NAMES

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
w = dataclasses.replace(syn.WORKLOADS['C5'], T=T, B=B)
cfg = w.model_config()
dev = torch.device('cuda:0')
p = {k: torch.tensor(v, device=dev) for k, v in syn.make_params(w, seed=1).items()}
g = torch.Generator(device=dev); g.manual_seed(3)
u = torch.randn(w.B, w.T, w.dim_u, dtype=torch.float64, device=dev, generator=g)
y = torch.randn(w.B, w.T, w.dim_y, dtype=torch.float64, device=dev, generator=g)
N = w.N
noise = {'hid_b': torch.randn(2 * T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_b': torch.randn(2 * T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_f': torch.randn((T - 1) * N, dtype=torch.float64, device=dev, generator=g)}
res = {}
for mode in ('recompute', 'tiles'):
    if mode == 'recompute':
        os.environ['CBFSSM_F32_NO_TILES'] = '1'
    else:
        os.environ.pop('CBFSSM_F32_NO_TILES', None)
    eng = HipElboGrad(cfg, dev, dtype='float32')
    loss, grads, _ = eng.loss_and_grads(p, u, y, noise)
    torch.cuda.synchronize()
    ws = eng.last_ws
    nb = 0 if ws.a2s_b is None else ws.a2s_b.numel() * 2
    print(mode, 'loss %.10g' % float(loss), 'record floats (backward runs): %d (2^31 = %d)' % (nb, 2 ** 31), flush=True)
    res[mode] = (float(loss), {k: grads[k].clone() for k in PARAM_NAMES})
    del eng
    torch.cuda.empty_cache()
worst = max(float((res['tiles'][1][k] - res['recompute'][1][k]).abs().max() / (res['recompute'][1][k].abs().max() + 1e-300)) for k in PARAM_NAMES)
print('loss difference %.3e, worst gradient difference (relative to the largest entry) %.3e' %
      (abs(res['tiles'][0] - res['recompute'][0]) / abs(res['recompute'][0]), worst))

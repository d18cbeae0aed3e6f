"""Diagnostic: cycles per adjoint step of one workgroup at full and at partial chip load (stamps build, CBFSSM_HIP_LIB=...):
if a step gets much cheaper when few workgroups run, the memory system (L2 latency under load) paces the kernel, not the CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import dataclasses
import torch
from cbfssm import synthetic as syn
from cbfssm.hip import train

name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
dev = 'cuda:0'
for B in (int(b) for b in (sys.argv[2:] or ['256', '64', '16'])):
    w = dataclasses.replace(syn.WORKLOADS[name], B=B) if dataclasses.is_dataclass(syn.WORKLOADS[name]) else None
    if w is None:
        w = syn.WORKLOADS[name]._replace(B=B)
    eng = train.HipElboGrad(w.model_config(), dev)
    params = {k: torch.tensor(v, device=dev) for k, v in syn.make_params(w).items()}
    g = torch.Generator(device=dev); g.manual_seed(0)
    u = torch.randn(w.B, w.T, w.dim_u, dtype=torch.float64, device=dev, generator=g)
    y = torch.randn(w.B, w.T, w.dim_y, dtype=torch.float64, device=dev, generator=g)
    N = w.B * w.S
    noise = {'hid_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
             'eps_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
             'eps_f': torch.randn((w.T - 1) * N, dtype=torch.float64, device=dev, generator=g)}
    reps = 4
    for _ in range(reps):
        eng.loss_and_grads(params, u, y, noise)
    torch.cuda.synchronize()
    ws = eng.last_ws
    groups = (N + 15) // 16
    for tag, slab, steps in (('fwd-adjoint', eng.red[:eng.slab_f], (w.T - 1) * groups), ('bwd-adjoint', eng.red[eng.slab_f:eng.slab_f + eng.slab_b], 2 * w.T * groups)):
        small = slab[-192:].cpu().numpy()
        tot = small[100:114].sum()
        print('%s B=%d (%d chain groups): %s %.0f cycles per workgroup-step (wave 0)' % (name, B, groups, tag, tot / steps))

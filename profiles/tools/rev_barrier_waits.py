"""Diagnostic: which wave is last at which barrier of the adjoint step -- barrier-wait shares of every row-block wave from the
-DCBF_REV_STAMPS -DCBF_STAMP_ALLWAVES build (make -C cbf-ssm_amd/csrc BUILD=build_stamps EXTRA="-DCBF_REV_STAMPS
-DCBF_STAMP_ALLWAVES" OUT=../lib/libcbfssm_hip_stamps.so; CBFSSM_HIP_LIB=... python profiles/tools/rev_barrier_waits.py C3).
The wave with the smallest wait at a barrier is the one the others waited for.  Never quote this build's run time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import torch
from cbfssm import synthetic as syn
from cbfssm.hip import train

w = syn.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'C3']
dev = 'cuda:0'
eng = train.HipElboGrad(w.model_config(), dev)
params = {k: torch.tensor(v, device=dev) for k, v in syn.make_params(w).items()}
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randn(w.B, w.T, w.dim_u, dtype=torch.float64, device=dev, generator=g)
y = torch.randn(w.B, w.T, w.dim_y, dtype=torch.float64, device=dev, generator=g)
N = w.N
noise = {'hid_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_f': torch.randn((w.T - 1) * N, dtype=torch.float64, device=dev, generator=g)}
for _ in range(10):
    eng.loss_and_grads(params, u, y, noise)
torch.cuda.synchronize()
names = ['0 (unused)', '1 K tile complete', '2 -', '3 -', '4 A2bar rows written (end of E)', '5 partial tiles written (end of F)', '6 end of G / D']
for tag, slab in (('forward-pass adjoint', eng.red[:eng.slab_f]), ('backward-run adjoint', eng.red[eng.slab_f:eng.slab_f + eng.slab_b])):
    small = slab[-192:].cpu().numpy()
    print(tag)
    tot = [small[100 + 12 * wv:100 + 12 * wv + 7].sum() + small[100 + 12 * wv + 7] for wv in range(7)]
    print('  barrier                                  ' + ''.join('  wave %d' % wv for wv in range(7)))
    for i in (1, 4, 5, 6):
        print('  %-40s' % names[i] + ''.join('  %5.1f%%' % (100 * small[100 + 12 * wv + i] / tot[wv]) for wv in range(7)))
    print('  %-40s' % 'all barriers' + ''.join('  %5.1f%%' % (100 * small[100 + 12 * wv:100 + 12 * wv + 7].sum() / tot[wv]) for wv in range(7)))

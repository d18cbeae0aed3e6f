"""Diagnostic: per-phase cycle shares of the adjoint kernels from the -DCBF_REV_STAMPS build
(make -C cbf-ssm_amd/csrc BUILD=build_stamps EXTRA=-DCBF_REV_STAMPS OUT=../lib/libcbfssm_hip_stamps.so).
Run with CBFSSM_HIP_LIB=cbf-ssm_amd/lib/libcbfssm_hip_stamps.so.  Never quote this build's run time."""
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
for _ in range(int(os.environ.get('REPS', '40'))):
    eng.loss_and_grads(params, u, y, noise)
torch.cuda.synchronize()
ws = eng.last_ws
names = ['-', 'B ktile (+ next inputs -> LDS)', '-', '-', 'E a2bar,gB', 'F kbar,xp,gZ', 'G carry + D(next)']
for tag, slab, nwg, nstep_total in (('fwd-adjoint', eng.red[:eng.slab_f], ws.n_f, (w.T - 1)),
                                    ('bwd-adjoint', eng.red[eng.slab_f:eng.slab_f + eng.slab_b], ws.n_b, None)):
    small = slab[-192:].cpu().numpy()
    print(tag, 'workgroups', nwg)
    print('  in-kernel clock: %.0f MHz (sum over workgroups of cycles / realtime ticks x 100 MHz)' % (100.0 * small[140] / max(small[141], 1.0)))
    for wname, o in (('wave0', 100), ('lastwave', 114)):
        c, wt = small[o:o + 7], small[o + 7:o + 14]
        tot = c.sum() + wt.sum()
        print('  %-8s total cycles (sum over WGs) %.4g' % (wname, tot))
        for i in range(7):
            print('     %-32s compute %5.1f%%  barrier-wait %5.1f%%' % (names[i], 100 * c[i] / tot, 100 * wt[i] / tot))
        if wname == 'wave0':
            marks = small[128:140]
            mn = ['C loop (only without saved A2)', '-', 'B rest (after the input loads)', 'D epilogue adjoint (next step)', '-', 'E whole',
                  'F loop(25)', 'F xp(8)', 'F transpose+gZ(8)', 'B saved rows -> LDS', 'B next inputs issued', 'loop back edge']
            if os.environ.get('MARKS_LAST'):
                print('        (sub-phase marks of the second recorded wave: -DCBF_STAMP_MARKS_LAST build)')
            for i in range(12):
                print('        sub %-20s %5.1f%%' % (mn[i], 100 * marks[i] / tot))

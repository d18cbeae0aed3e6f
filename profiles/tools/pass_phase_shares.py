"""Diagnostic: per-phase cycle shares of the forward-evaluation pass kernels from the -DCBF_REV_STAMPS build.
Run with CBFSSM_HIP_LIB=cbf-ssm_amd/lib/libcbfssm_hip_stamps.so CBFSSM_NC_FWD=1 CBFSSM_NC_BWD=1."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import torch
from cbfssm import synthetic as syn
from cbfssm.hip import ops, lib

w = syn.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'C3']
dev = 'cuda:0'
l = lib.load()
eng = ops.HipElbo(w.model_config(), dev)
eng.prepare({k: torch.tensor(v, device=dev) for k, v in syn.make_params(w).items()})
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randn(w.B, w.T, w.dim_u, dtype=torch.float64, device=dev, generator=g)
y = torch.randn(w.B, w.T, w.dim_y, dtype=torch.float64, device=dev, generator=g)
N = w.N
noise = {'hid_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_b': torch.randn(2 * w.T * N, dtype=torch.float64, device=dev, generator=g),
         'eps_f': torch.randn((w.T - 1) * N, dtype=torch.float64, device=dev, generator=g)}
prob = eng.problem(w.B, w.T, True)
ws = ops.ElboWorkspace(prob, dev)
dbg = torch.zeros(64 * 8192, dtype=torch.float64, device=dev)
l.cbfssm_debug_set_buffer.argtypes = [C.c_void_p]
l.cbfssm_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
st = ops._stream()
names = ['phase3 + loads', 'phase1 (tile,exp)', 'phase2 (K^-1 K, P)']
for tag in ('bwd', 'fwd'):
    dbg.zero_()
    for _ in range(5):
        if tag == 'bwd':
            lib.check(l.cbfssm_backward_pass_f64(C.byref(prob), C.byref(eng.pack_b.layout), ops._ptr(eng.pack_b.buf),
                      ops._ptr(eng.var_x), ops._ptr(u), ops._ptr(y), ops._ptr(noise['hid_b']), ops._ptr(noise['eps_b']),
                      ops._ptr(ws.y2), None, None, None, ops._ptr(ws.ent_part), st), 'bwd')
        else:
            lib.check(l.cbfssm_forward_pass_f64(C.byref(prob), C.byref(eng.pack_f.layout), ops._ptr(eng.pack_f.buf),
                      ops._ptr(eng.var_x), ops._ptr(eng.var_y), ops._ptr(u), ops._ptr(y), ops._ptr(ws.y2),
                      ops._ptr(noise['eps_f']), ops._ptr(ws.x), None, None, ops._ptr(ws.kl_part), st), 'fwd')
    torch.cuda.synchronize()
    d = dbg.view(-1, 64).cpu().numpy()
    d = d[d.sum(1) > 0]
    print(tag, 'workgroups', d.shape[0])
    for wname, o in (('wave0', 0), ('lastwave', 32)):
        c, wt = d[:, o:o + 3].sum(0), d[:, o + 7:o + 10].sum(0)
        tot = c.sum() + wt.sum()
        print('  %-8s cycles per workgroup %.4g' % (wname, tot / d.shape[0]))
        for i in range(3):
            print('     %-22s compute %5.1f%%  barrier-wait %5.1f%%' % (names[i], 100 * c[i] / tot, 100 * wt[i] / tot))

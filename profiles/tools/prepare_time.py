import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import torch
from cbfssm import synthetic as syn
from cbfssm.hip import ops
for name in ('C1','C2','C3','C4','C5'):
    w = syn.WORKLOADS[name]
    eng = ops.HipElbo(w.model_config(), 'cuda:0')
    p = {k: torch.tensor(v, device='cuda:0') for k, v in syn.make_params(w).items()}
    for _ in range(3): eng.prepare(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.prepare(p)
    e1.record(); e1.synchronize()
    print(name, 'M', w.M, 'prepare (both GPs, incl. softplus glue): %.1f us' % (e0.elapsed_time(e1) * 1e3 / 20))

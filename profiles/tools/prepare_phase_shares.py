import os, sys
ROOT='/root/repo'; sys.path[:0]=[ROOT, os.path.join(ROOT,'cbf-ssm_amd')]
import torch
from cbfssm import synthetic as syn
from cbfssm.hip import ops
for name in ('C3','C4','C5'):
    w = syn.WORKLOADS[name]
    eng = ops.HipElbo(w.model_config(), 'cuda:0')
    p = {k: torch.tensor(v, device='cuda:0') for k, v in syn.make_params(w).items()}
    eng.prepare(p); torch.cuda.synchronize()
    s = eng.pack_f.scal.cpu().numpy()
    print(name, 'M', w.M, 'cycles: panels %.0f  updates %.0f  outputs+Kinv %.0f  images+KL %.0f  (us at 2.4GHz: %.0f %.0f %.0f %.0f)' % (s[4],s[5],s[6],s[7], s[4]/2400, s[5]/2400, s[6]/2400, s[7]/2400))

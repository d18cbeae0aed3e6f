import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cbf-ssm_amd')]
import torch
from cbfssm.hip import lib, ops
dev = 'cuda:0'
n = 5 * 250 * 5120 - 5120
buf = torch.empty(n, dtype=torch.float64, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
l = lib.load()
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('elements %d (%.1f MB)' % (n, n * 8 / 1e6))
print('cbfssm_normal_f64: %.1f us' % t(lambda: l.cbfssm_normal_f64(1, 0, n, ops._ptr(buf), ops._stream())))
print('torch normal_ (Philox, float64): %.1f us' % t(lambda: buf.normal_(generator=g)))

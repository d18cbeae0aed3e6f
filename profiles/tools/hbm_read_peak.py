"""Context for roofline.hbm_kernel: what a plain streaming read of the same 143 MB reaches on this box (buffers rotated so that every
read comes from HBM, as bench.py does): the tensor library's sum reduction and a float64 copy."""
import torch
dev = 'cuda:0'
n = 250 * 5120 * 14
bufs = [torch.randn(n, dtype=torch.float64, device=dev) for _ in range(6)]
out = torch.empty(n, dtype=torch.float64, device=dev)
def t(fn, reps=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
ts = t(lambda i: bufs[i % 6].sum())
print('sum over %.0f MB: %.1f us = %.2f TB/s (%.2f of 8)' % (n * 8 / 1e6, ts * 1e6, n * 8 / ts / 1e12, n * 8 / ts / 8e12))
tc = t(lambda i: out.copy_(bufs[i % 6]))
print('copy (read + write %.0f MB): %.1f us = %.2f TB/s (%.2f of 8)' % (2 * n * 8 / 1e6, tc * 1e6, 2 * n * 8 / tc / 1e12, 2 * n * 8 / tc / 8e12))

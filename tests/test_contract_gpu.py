"""The stash-mode contraction through the C ABI (cbfssm_stash_contract_f64) on random operand images: the symmetric form
(default: the lower-triangular blocks of sum A2bar K^T + K A2bar^T, expanded to the symmetric part of the sum -- all that
d loss / d K_mm = -K^-1 (.) K^-1 uses) and the full-matrix forms (CBFSSM_CONTRACT_FULL=1) against a float64 matrix product,
at the four tile heights that run in stash mode, with slot counts that do not divide the slice count."""
import ctypes as C

import numpy as np
import pytest
import torch

from cbfssm.hip import lib, ops
from cbfssm.hip.train import _unpack_c

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _dense(img, nblk, nslots):
    """operand image [slot][rb][s][lane] -> X (16 nblk, 16 nslots): X[16 rb + (lane & 15)][16 slot + 4 s + (lane >> 4)]"""
    v = img.view(nslots, nblk, 4, 4, 16)                    # slot, rb, s, g = lane >> 4, m = lane & 15
    return v.permute(1, 4, 0, 2, 3).reshape(16 * nblk, 16 * nslots)


@pytest.mark.parametrize('M,nslots', [(150, 5), (200, 37), (200, 1000), (250, 19), (300, 23)])
def test_contraction_matches_a_matrix_product(M, nslots, monkeypatch):
    l = lib.load()
    lay = lib.pack_layout(M, 6, 4)
    nblk = lay.NBLK
    assert lay.rev_stash == 1
    g = torch.Generator(device=DEV)
    g.manual_seed(M + nslots)
    sa = torch.randn(nslots * nblk * 256, dtype=torch.float64, device=DEV, generator=g)
    sk = torch.randn(nslots * nblk * 256, dtype=torch.float64, device=DEV, generator=g)
    B = _dense(sa, nblk, nslots) @ _dense(sk, nblk, nslots).T
    work = torch.zeros(int(l.cbfssm_stash_contract_work_elems(C.byref(lay), nslots)), dtype=torch.float64, device=DEV)
    for full, want in ((False, 0.5 * (B + B.T)), (True, B)):
        if full:
            monkeypatch.setenv('CBFSSM_CONTRACT_FULL', '1')
        img = torch.full((nblk * nblk * 256,), 1.0, dtype=torch.float64, device=DEV)      # the call ADDS to the image
        for _ in range(2):
            lib.check(l.cbfssm_stash_contract_f64(C.byref(lay), ops._ptr(sa), ops._ptr(sk), nslots, ops._ptr(work),
                                                  ops._ptr(img), ops._stream()), 'contract')
        torch.cuda.synchronize()
        got = _unpack_c(img, nblk, nblk)
        err = float((got - (1.0 + 2.0 * want)).abs().max()) / float(want.abs().max())
        assert err < 1e-13, (M, nslots, full, err)
        if not full:
            assert float((got - got.T).abs().max()) == 0.0                                # symmetric to the last bit

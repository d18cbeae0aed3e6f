"""cbfssm_normal_f64 (the library's stand-in for the reference's in-graph tf.random_normal, cbfssm.py:134,149,209) against its
numpy restatement, which the Philox paper's known-answer vectors pin (tests/test_oracle.py::test_philox_known_answers)."""
import ctypes as C
import numpy as np
import pytest
import torch

from cbfssm.hip import lib, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _normal(seed, offset, n):
    out = torch.full((n + 2,), 7.0, dtype=torch.float64, device=DEV)                   # guards on both sides
    lib.check(lib.load().cbfssm_normal_f64(seed, offset, n, ops._ptr(out[1:]), ops._stream()), 'normal')
    o = out.cpu().numpy()
    assert o[0] == 7.0 and o[-1] == 7.0
    return o[1:-1]


def test_generator_words_and_known_answer():
    from oracle import philox as ph
    l = lib.load()
    for seed, first, n in ((0, 0, 4), (0xa4093822299f31d0, 2 ** 32 - 2, 70), (2 ** 64 - 1, 2 ** 40 + 5, 1000)):
        out = torch.zeros(4 * n, dtype=torch.int32, device=DEV)
        lib.check(l.cbfssm_philox4x32_10_u32(seed, first, n, C.c_void_p(out.data_ptr()), ops._stream()), 'philox')
        got = out.cpu().numpy().view(np.uint32).reshape(n, 4)
        p = np.uint64(first) + np.arange(n, dtype=np.uint64)
        ctr = np.stack([p & np.uint64(0xFFFFFFFF), p >> np.uint64(32), 0 * p, 0 * p], axis=-1)
        key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64), (n, 2))
        np.testing.assert_array_equal(got, ph.philox4x32_10(ctr, key))
        if seed == 0 and first == 0:
            assert [int(v) for v in got[0]] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]   # the paper's vector


@pytest.mark.parametrize('seed,offset,n', [(1, 0, 1), (1, 1, 1), (7, 0, 1000), (7, 3, 1001), (2 ** 63 + 11, 2 ** 33 + 1, 4097),
                                            (5, 0, 3_000_000)])
def test_normal_matches_restatement(seed, offset, n):
    from oracle import philox as ph
    got, want = _normal(seed, offset, n), ph.normal(seed, offset, n)
    # (the device's log / sincospi and numpy's differ in the last bits: a few ulps of values of order one)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)


def test_split_draws_and_pipeline_reproducibility():
    whole = _normal(99, 10, 5000)
    np.testing.assert_array_equal(np.concatenate([_normal(99, 10, 1233), _normal(99, 1243, 3767)]), whole)
    assert lib.load().cbfssm_normal_f64(1, 0, 0, None, ops._stream()) == 0             # nothing to draw
    with pytest.raises(lib.CbfssmHipError):
        lib.check(lib.load().cbfssm_normal_f64(1, 0, 5, None, ops._stream()), 'normal')

    def run(seed):
        g = torch.Generator(device=DEV)
        g.manual_seed(seed)
        pipe = ops.NoisePipeline(DEV, g)
        return [{k: v.clone() for k, v in pipe.next(T, N).items()} for T, N in ((5, 8), (9, 8), (5, 8), (5, 8))]
    a, b, c = run(3), run(3), run(4)
    for x, y, z in zip(a, b, c):
        assert set(x) == {'hid_b', 'eps_b', 'eps_f'}
        for k in x:
            assert torch.equal(x[k], y[k]) and not torch.equal(x[k], z[k])
    assert not torch.equal(a[0]['eps_f'], a[2]['eps_f'])                                 # a new draw every call
    flat = torch.cat([v for d in a for v in (d['hid_b'], d['eps_b'], d['eps_f'])]).cpu().numpy()
    assert abs(flat.mean()) < 0.1 and abs(flat.std() - 1.0) < 0.1

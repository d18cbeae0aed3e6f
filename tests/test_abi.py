"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/cbfssm_hip.h declares.
(No compute calls here: those need a GPU and live in test_hip_parity.py.)"""
import ctypes
import os
import re

from cbfssm.hip import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cbfssm_hip.h')).read()
    return sorted(set(re.findall(r'\b(cbfssm_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), 'run __graft_entry__.build() first'
    so = ctypes.CDLL(lib.LIB_PATH)
    declared = _declared_symbols()
    assert set(declared) == set(lib.SYMBOLS)
    for name in declared:
        assert hasattr(so, name), name


def test_layout_is_host_only_and_consistent():
    lay = lib.pack_layout(100, 21, 14)
    assert (lay.NBLK, lay.DK, lay.Mp, lay.KS) == (7, 6, 112, 28)
    assert lay.total > lay.Zs >= lay.Linvt + 100 * 100
    lay = lib.pack_layout(300, 6, 4)
    assert (lay.NBLK, lay.DK) == (20, 2)
    for bad in ((0, 5, 4), (321, 5, 4), (10, 25, 4), (10, 5, 17)):
        try:
            lib.pack_layout(*bad)
            assert False, bad
        except lib.CbfssmHipError:
            pass
    assert lib.load().cbfssm_version() >= 1


def test_param_layout_is_host_only_and_ordered():
    """cbfssm_param_layout_init: the twelve tensors of CBFSSM._setup_vars (reference cbfssm.py:30-58) in PARAM order."""
    pl = lib.param_layout(100, 14, 7, 7)
    M, D, dx, dob = 100, 21, 14, 7
    sizes = [M * D, M * dx, M * dx, 1, D, M * D, M * dob, M * dob, 1, D, dx, dx]
    off = 0
    for k, n in enumerate(sizes):
        assert pl.off[k] == off, k
        off += n
    assert pl.total == off and (pl.M, pl.D, pl.dim_x, pl.dim_y) == (M, D, dx, 7)
    for bad in ((0, 4, 1, 1), (10, 4, 1, 5), (10, 0, 1, 1)):
        try:
            lib.param_layout(*bad)
            assert False, bad
        except lib.CbfssmHipError:
            pass


def test_product_never_imports_the_oracle_and_has_no_fallback(tmp_path, monkeypatch):
    """oracle/ is test infrastructure: no module of the package (nor a profiling tool) may import it, and a missing
    shared library is an error, not a reason to compute somewhere else."""
    import ast
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for top in ('cbf-ssm_amd', os.path.join('profiles', 'tools')):
        for d, _, files in os.walk(os.path.join(root, top)):
            for f in files:
                if not f.endswith('.py'):
                    continue
                tree = ast.parse(open(os.path.join(d, f)).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or '']
                    if any(n == 'oracle' or n.startswith('oracle.') for n in names):
                        offenders.append(os.path.join(d, f))
    assert not offenders, offenders
    code = ('import sys; sys.path[:0] = [%r, %r]\n'
            'from cbfssm.hip import lib\n'
            'try:\n    lib.load()\nexcept lib.CbfssmHipError as e:\n    print("LOUD:", e)\n'
            % (root, os.path.join(root, 'cbf-ssm_amd')))
    env = dict(os.environ, CBFSSM_HIP_LIB=str(tmp_path / 'absent.so'))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=240)
    assert 'LOUD:' in out.stdout and 'no fallback' in out.stdout, (out.stdout, out.stderr)


def test_element_counts_come_back_as_64_bit():
    """Every `*_elems` entry point returns int64_t and the binding says so: at the C5 shape the float32 record buffer of the
    backward runs has 3.3e10 floats -- a 32-bit result wraps, the buffer is allocated far too small and the passes write
    outside it (found in round 4: a GPU memory fault above 2^31 floats).  Host arithmetic only: no GPU needed."""
    l = lib.load()
    for name in lib.SYMBOLS:
        if name.endswith('_elems'):
            assert getattr(l, name).restype is ctypes.c_int64, name
    prob = lib.make_problem(512, 50, 1000, 4, 2, 2, 300, 50, 1.0, True)
    lay = lib.pack_layout(300, 6, 2)
    n32 = l.cbfssm_saved_a2_f32_elems(ctypes.byref(prob), ctypes.byref(lay), 1)
    assert n32 == 2 * 1000 * 1600 * 2 * 20 * 256 and n32 > 2 ** 31
    n64 = l.cbfssm_saved_a2_elems(ctypes.byref(prob), ctypes.byref(lay), 1)
    assert n64 == 2 * 1000 * 1600 * 20 * 256


def test_variant_entry_points_reject_the_other_variants_problem():
    """The forward-only variants and CBFSSM share kernels but not entry points: a problem with half = 1 handed to a CBFSSM
    entry point (or the other way round) is an error with a message, decided on the host before anything is launched --
    float64 and float32 alike -- and so is an unknown g_mode of the variants' train tail."""
    import ctypes as C
    l = lib.load()
    lay = lib.pack_layout(20, 5, 4)
    lay_b = lib.pack_layout(20, 5, 2)
    full = lib.make_problem(2, 3, 6, 4, 1, 2, 20, 3, 1.0, True)
    half = lib.make_problem(2, 3, 6, 4, 1, 2, 20, 3, 1.0, True, half=True)
    nul = [None] * 16

    def rc(fn, prob, layout, nargs, *tail):
        return fn(C.byref(prob), C.byref(layout), *nul[:nargs], *tail)
    cases = [
        (l.cbfssm_forward_pass_f64, half, lay, 12), (l.cbfssm_half_forward_pass_f64, full, lay, 12),
        (l.cbfssm_forward_pass_f32, half, lay, 12), (l.cbfssm_half_forward_pass_f32, full, lay, 12),
        (l.cbfssm_backward_pass_f32, half, lay_b, 12),
    ]
    for fn, prob, layout, nargs in cases:
        assert rc(fn, prob, layout, nargs) != 0, fn.__name__
        assert l.cbfssm_last_error().decode(), fn.__name__
    assert l.cbfssm_forward_pass_bwd_f32(C.byref(half), C.byref(lay), *nul[:10], 1.0, None, None, None) != 0
    assert b'half' in l.cbfssm_last_error()
    assert l.cbfssm_half_forward_pass_bwd_f32(C.byref(full), C.byref(lay), *nul[:9], 1.0, None, None, None) != 0
    assert b'half' in l.cbfssm_last_error()
    assert l.cbfssm_train_tail_half_f64(C.byref(lay), None, None, 0, None, None, 0, 3, 2, None, None, None, None, None) != 0
    assert b'g_mode' in l.cbfssm_last_error()

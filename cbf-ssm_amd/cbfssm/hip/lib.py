"""ctypes binding of libcbfssm_hip.so (C ABI: include/cbfssm_hip.h).

The library is the product: if it cannot be loaded every entry point raises -- there is no CPU or eager-PyTorch
fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CBFSSM_HIP_LIB') or os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libcbfssm_hip.so'))

SCAL_SIGMA2, SCAL_LOGDET, SCAL_KLZ, SCAL_INFO, SCAL_COND, SCAL_JITTER, SCAL_COUNT = 0, 1, 2, 3, 4, 5, 16
GP_FORM_DENSE, GP_FORM_TRI = 0, 1
JITTER = 1e-8   # cbfssm/model/gp_tf.py:57

# every symbol include/cbfssm_hip.h declares (tests check the shared object exports all of them)
SYMBOLS = (
    'cbfssm_last_error', 'cbfssm_version', 'cbfssm_gp_pack_layout', 'cbfssm_kmm_chol_f64', 'cbfssm_gp_prepare_f64',
    'cbfssm_gp_predict_f64', 'cbfssm_backward_pass_partials', 'cbfssm_backward_pass_f64',
    'cbfssm_forward_pass_partials', 'cbfssm_forward_pass_f64', 'cbfssm_loglik_moments_f64',
    'cbfssm_elbo_combine_f64', 'cbfssm_rev_workgroups', 'cbfssm_forward_pass_bwd_f64',
    'cbfssm_backward_pass_bwd_f64', 'cbfssm_reduce_partials_f64', 'cbfssm_gp_prepare2_f64', 'cbfssm_bwd_segments', 'cbfssm_forward_pass_bwd_ex_f64',
    'cbfssm_backward_pass_bwd_ex_f64', 'cbfssm_half_forward_pass_f64', 'cbfssm_half_forward_pass_bwd_f64',
    'cbfssm_saved_a2_elems', 'cbfssm_param_layout_init', 'cbfssm_constrain_f64', 'cbfssm_train_tail_work_elems',
    'cbfssm_train_tail_f64', 'cbfssm_train_tail_g_f64', 'cbfssm_train_tail_half_work_elems', 'cbfssm_train_tail_half_f64', 'cbfssm_gru_recog_param_elems', 'cbfssm_gru_recog_act_elems', 'cbfssm_gru_recog_f64', 'cbfssm_gru_recog_bwd_f64', 'cbfssm_adam_step_f64', 'cbfssm_loglik_partials', 'cbfssm_data_tail_f64', 'cbfssm_stash_contract_work_elems', 'cbfssm_stash_contract_f64',
    'cbfssm_cholesky_f64', 'cbfssm_rbf_k_f64', 'cbfssm_gp_predict_fullq_work_elems', 'cbfssm_gp_predict_fullq_f64',
    'cbfssm_pack_f32_elems', 'cbfssm_gp_pack_f32', 'cbfssm_gp_pack_bf16', 'cbfssm_gp_predict_f32', 'cbfssm_backward_pass_f32', 'cbfssm_forward_pass_f32',
    'cbfssm_saved_a2_f32_elems', 'cbfssm_half_forward_pass_f32', 'cbfssm_half_forward_pass_bwd_f32', 'cbfssm_rev32_slab_elems', 'cbfssm_normal_f64', 'cbfssm_philox4x32_10_u32', 'cbfssm_forward_pass_bwd_f32', 'cbfssm_backward_pass_bwd_f32',
)


class ParamLayout(C.Structure):
    _fields_ = [('off', C.c_int64 * 12), ('total', C.c_int64)] + [(n, C.c_int32) for n in ('M', 'D', 'dim_x', 'dim_y')]


class PackLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('total', 'Bp', 'Zp', 'cz', 'muA', 's2A', 'invl', 'scal', 'Kmm', 'L', 'Kinv',
                                          'Linvt', 'Zs', 'muB', 's2B', 'ZT', 'rev_slab', 'work', 'Wp', 'WTp')] + \
               [(n, C.c_int32) for n in ('M', 'D', 'Do', 'NBLK', 'DK', 'Mp', 'Dp', 'KS', 'JB', 'rev_stash', 'gp_form',
                                         'reserved')]


class Problem(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('B', 'S', 'T', 'dim_x', 'dim_u', 'dim_y', 'M', 'recog_len', 'condition',
                                          'half', 'group0', 'ngroups')] + [('k_factor', C.c_double)]


class CbfssmHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises if it is missing: build it with `python __graft_entry__.py`."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch owns the device memory these entry points work on: its HIP runtime (the libamdhip64 it bundles) must be
    # the one this process binds, so it is loaded first -- otherwise the library would resolve the system runtime and
    # the two would not share a context ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CbfssmHipError('HIP extension not built: %s is missing (run __graft_entry__.build() / make -C '
                             'cbf-ssm_amd/csrc). There is no fallback path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.cbfssm_last_error.restype = C.c_char_p
    lib.cbfssm_last_error.argtypes = []
    lib.cbfssm_version.restype = ip
    lib.cbfssm_gp_pack_layout.argtypes = [ip, ip, ip, C.POINTER(PackLayout)]
    lib.cbfssm_kmm_chol_f64.argtypes = [ip, ip, vp, vp, vp, dbl, vp, vp, vp, vp, vp]
    lib.cbfssm_gp_prepare_f64.argtypes = [C.POINTER(PackLayout), vp, vp, vp, vp, vp, dbl, vp, vp]
    lib.cbfssm_gp_prepare2_f64.argtypes = ([C.POINTER(PackLayout)] + [vp] * 6) * 2 + [dbl, vp]
    lib.cbfssm_gp_predict_f64.argtypes = [C.POINTER(PackLayout), vp, vp, i64, vp, vp, vp]
    lib.cbfssm_backward_pass_partials.restype = i64
    lib.cbfssm_backward_pass_partials.argtypes = [C.POINTER(Problem)]
    lib.cbfssm_backward_pass_f64.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_forward_pass_partials.restype = i64
    lib.cbfssm_loglik_partials.restype = i64
    lib.cbfssm_loglik_partials.argtypes = [C.POINTER(Problem)]
    lib.cbfssm_param_layout_init.argtypes = [ip, ip, ip, ip, C.POINTER(ParamLayout)]
    lib.cbfssm_constrain_f64.argtypes = [C.POINTER(ParamLayout), vp, vp, vp]
    lib.cbfssm_train_tail_work_elems.restype = i64
    lib.cbfssm_train_tail_work_elems.argtypes = [C.POINTER(PackLayout), C.POINTER(PackLayout)]
    lib.cbfssm_train_tail_f64.argtypes = ([C.POINTER(ParamLayout), C.POINTER(PackLayout), vp, C.POINTER(PackLayout), vp, vp,
                                          vp, vp, i64, vp, vp, vp, vp, vp])
    lib.cbfssm_train_tail_g_f64.argtypes = ([C.POINTER(ParamLayout), C.POINTER(PackLayout), vp, C.POINTER(PackLayout), vp, vp,
                                            vp, vp, i64, ip, vp, vp, vp, vp, vp])
    lib.cbfssm_train_tail_half_work_elems.restype = i64
    lib.cbfssm_train_tail_half_work_elems.argtypes = [C.POINTER(PackLayout)]
    lib.cbfssm_train_tail_half_f64.argtypes = [C.POINTER(PackLayout), vp, vp, ip, vp, vp, i64, ip, ip, vp, vp, vp, vp, vp]
    lib.cbfssm_gru_recog_param_elems.restype = i64
    lib.cbfssm_gru_recog_param_elems.argtypes = [ip, ip, ip]
    lib.cbfssm_gru_recog_act_elems.restype = i64
    lib.cbfssm_gru_recog_act_elems.argtypes = [ip, ip]
    lib.cbfssm_gru_recog_f64.argtypes = [ip, ip, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp]
    lib.cbfssm_gru_recog_bwd_f64.argtypes = [ip, ip, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp]
    lib.cbfssm_stash_contract_work_elems.restype = i64
    lib.cbfssm_stash_contract_work_elems.argtypes = [C.POINTER(PackLayout), i64]
    lib.cbfssm_stash_contract_f64.argtypes = [C.POINTER(PackLayout), vp, vp, i64, vp, vp, vp]
    lib.cbfssm_data_tail_f64.argtypes = [C.POINTER(Problem), vp, vp, vp, dbl, vp, vp]
    lib.cbfssm_adam_step_f64.argtypes = [i64, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, vp]
    lib.cbfssm_saved_a2_elems.restype = i64
    lib.cbfssm_saved_a2_elems.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout), ip]
    lib.cbfssm_forward_pass_partials.argtypes = [C.POINTER(Problem)]
    lib.cbfssm_forward_pass_f64.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_loglik_moments_f64.argtypes = [C.POINTER(Problem)] + [vp] * 9
    lib.cbfssm_elbo_combine_f64.argtypes = [C.POINTER(Problem), dbl, dbl, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp]
    lib.cbfssm_rev_workgroups.restype = i64
    lib.cbfssm_rev_workgroups.argtypes = [C.POINTER(Problem), ip]
    lib.cbfssm_forward_pass_bwd_f64.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, vp, vp]
    lib.cbfssm_backward_pass_bwd_f64.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, vp]
    lib.cbfssm_bwd_segments.argtypes = [C.POINTER(Problem)]
    lib.cbfssm_forward_pass_bwd_ex_f64.argtypes = ([C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, vp, ip, ip, vp, vp, vp, i64, vp])
    lib.cbfssm_backward_pass_bwd_ex_f64.argtypes = ([C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, ip, ip, ip, vp, vp, i64, vp])
    lib.cbfssm_half_forward_pass_f64.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_half_forward_pass_bwd_f64.argtypes = ([C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 9 + [dbl, vp, vp, ip, ip, vp, vp, vp, i64, vp])
    lib.cbfssm_reduce_partials_f64.argtypes = [vp, i64, i64, vp, vp]
    lib.cbfssm_cholesky_f64.argtypes = [ip, vp, dbl, vp, vp, vp, vp]
    lib.cbfssm_rbf_k_f64.argtypes = [ip, ip, ip, vp, vp, vp, vp, vp, vp]
    lib.cbfssm_gp_predict_fullq_work_elems.restype = i64
    lib.cbfssm_gp_predict_fullq_work_elems.argtypes = [C.POINTER(PackLayout), i64]
    lib.cbfssm_gp_predict_fullq_f64.argtypes = [C.POINTER(PackLayout), vp, vp, vp, i64, vp, vp, vp, vp]
    lib.cbfssm_pack_f32_elems.restype = i64
    lib.cbfssm_pack_f32_elems.argtypes = [C.POINTER(PackLayout)]
    lib.cbfssm_gp_pack_f32.argtypes = [C.POINTER(PackLayout), vp, vp, vp]
    lib.cbfssm_gp_pack_bf16.argtypes = [C.POINTER(PackLayout), vp, vp, vp]
    lib.cbfssm_gp_predict_f32.argtypes = [C.POINTER(PackLayout), vp, vp, i64, vp, vp, vp]
    lib.cbfssm_backward_pass_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_forward_pass_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_half_forward_pass_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 12
    lib.cbfssm_half_forward_pass_bwd_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 9 + [dbl, vp, vp, vp]
    lib.cbfssm_saved_a2_f32_elems.restype = i64
    lib.cbfssm_normal_f64.argtypes = [C.c_uint64, C.c_uint64, i64, vp, vp]
    lib.cbfssm_philox4x32_10_u32.argtypes = [C.c_uint64, C.c_uint64, i64, vp, vp]
    lib.cbfssm_saved_a2_f32_elems.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout), ip]
    lib.cbfssm_rev32_slab_elems.restype = i64
    lib.cbfssm_rev32_slab_elems.argtypes = [C.POINTER(PackLayout)]
    lib.cbfssm_forward_pass_bwd_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, vp, vp]
    lib.cbfssm_backward_pass_bwd_f32.argtypes = [C.POINTER(Problem), C.POINTER(PackLayout)] + [vp] * 10 + [dbl, vp, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name.endswith('_elems'):                 # element counts: 64-bit results (set above)
            assert fn.restype is i64, name
            continue
        if fn.restype is C.c_int or name.endswith('_f64') or name.endswith('_f32') or name.endswith('_bf16') or name in ('cbfssm_gp_pack_layout', 'cbfssm_bwd_segments',
                                                                        'cbfssm_param_layout_init'):
            fn.restype = ip
    _lib = lib
    return lib


def param_layout(M, dim_x, dim_u, dim_y):
    pl = ParamLayout()
    check(load().cbfssm_param_layout_init(int(M), int(dim_x), int(dim_u), int(dim_y), C.byref(pl)), 'cbfssm_param_layout_init')
    return pl


def check(rc, what):
    if rc != 0:
        msg = load().cbfssm_last_error().decode('utf-8', 'replace')
        raise CbfssmHipError('%s failed (rc=%d): %s' % (what, rc, msg))


def pack_layout(M, D, Do):
    lay = PackLayout()
    check(load().cbfssm_gp_pack_layout(int(M), int(D), int(Do), C.byref(lay)), 'cbfssm_gp_pack_layout')
    return lay


def make_problem(B, S, T, dim_x, dim_u, dim_y, M, recog_len, k_factor, condition, half=False):
    p = Problem()
    p.B, p.S, p.T, p.dim_x, p.dim_u, p.dim_y, p.M = int(B), int(S), int(T), int(dim_x), int(dim_u), int(dim_y), int(M)
    p.recog_len, p.condition, p.half, p.k_factor = int(recog_len), int(bool(condition)), int(bool(half)), float(k_factor)
    p.group0, p.ngroups = 0, 0
    return p

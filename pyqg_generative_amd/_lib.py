"""ctypes binding of the C-ABI library ``libqgx.so`` (include/qgx.h).

The HIP library is the product: there is NO CPU fallback.  Importing this
module without the built library raises ``ImportError`` loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QGX_LIB') or os.path.join(_HERE, 'libqgx.so')   # QGX_LIB: developer builds only


class QgxError(RuntimeError):
    pass


class qgx_config(C.Structure):
    _fields_ = [('nx', C.c_int32), ('n_members', C.c_int32), ('device', C.c_int32),
                ('plan_only', C.c_int32),
                ('L', C.c_double), ('dt', C.c_double), ('rek', C.c_double), ('delta', C.c_double),
                ('beta', C.c_double), ('rd', C.c_double), ('U1', C.c_double), ('U2', C.c_double),
                ('H1', C.c_double), ('filterfac', C.c_double)]


class qgx_param(C.Structure):
    _fields_ = [('gen', C.c_void_p), ('sampling', C.c_int32), ('nsteps', C.c_int32),
                ('weight', C.c_double), ('seed', C.c_uint64), ('member_offset', C.c_uint64),
                ('z_external_dev', C.c_void_p), ('forcing_dev', C.c_void_p),
                ('demean', C.c_int32), ('reserved', C.c_int32)]


class qgx_cnn_weights(C.Structure):
    _fields_ = [('n_in', C.c_int32), ('n_out', C.c_int32),
                ('conv_w', C.c_void_p * 8), ('conv_b', C.c_void_p * 8),
                ('bn_gamma', C.c_void_p * 7), ('bn_beta', C.c_void_p * 7),
                ('bn_mean', C.c_void_p * 7), ('bn_var', C.c_void_p * 7),
                ('bn_eps', C.c_float)]


# enum mirrors (include/qgx.h)
F_Q, F_QH, F_PH, F_U, F_V, F_DQHDT, F_DQHDT_P, F_DQHDT_PP, F_S, F_Z, F_P = range(11)
T_FILTR, T_WV2, T_A, T_KK, T_LL = range(5)
SAMPLING_AR1, SAMPLING_CONSTANT = 0, 1
DIAGS = ['KEspec', 'Ensspec', 'entspec', 'APEflux', 'KEflux', 'APEgenspec', 'KEfrictionspec', 'paramspec',
         'paramspec_APEflux', 'paramspec_KEflux', 'Dissspec', 'ENSDissspec', 'ENSflux', 'ENSgenspec', 'ENSfrictionspec',
         'ENSparamspec']
GEN_GAN, GEN_VAE, GEN_GZ = 0, 1, 2

# every symbol include/qgx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('qgx_create', C.c_int, [C.POINTER(qgx_config), C.POINTER(C.c_void_p)]),
    ('qgx_destroy', C.c_int, [C.c_void_p]),
    ('qgx_set_q', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('qgx_set_qh', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('qgx_get', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ('qgx_get_table', C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    ('qgx_field_bytes', C.c_size_t, [C.c_void_p, C.c_int]),
    ('qgx_invert', C.c_int, [C.c_void_p, C.c_void_p]),
    ('qgx_step', C.c_int, [C.c_void_p, C.c_int, C.POINTER(qgx_param), C.c_int, C.c_void_p]),
    ('qgx_step_streams', C.c_int, [C.c_void_p, C.POINTER(qgx_param)]),
    ('qgx_step_count', C.c_int64, [C.c_void_p]),
    ('qgx_run_kernel_state', C.c_int, [C.c_void_p]),
    ('qgx_reset_time', C.c_int, [C.c_void_p]),
    ('qgx_set_option', C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ('qgx_status_ke_cfl', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('qgx_diag_config', C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    ('qgx_diag_get', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ('qgx_diag_count', C.c_int64, [C.c_void_p]),
    ('qgx_diag_reset', C.c_int, [C.c_void_p]),
    ('qgx_generator_create', C.c_int, [C.c_int, C.POINTER(qgx_cnn_weights), C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                       C.POINTER(C.c_void_p)]),
    ('qgx_generator_destroy', C.c_int, [C.c_void_p]),
    ('qgx_generator_forward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]),
    ('qgx_cnn_forward', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_void_p]),
    ('qgx_rfft2', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('qgx_irfft2', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('qgx_spec_regrid', C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p]),
    ('qgx_spec_div', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    ('qgx_real_fma', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p,
                               C.c_double, C.c_void_p]),
    ('qgx_moments_accumulate', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ('qgx_generator_set_option', C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ('qgx_generator_range_read', C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_float), C.c_void_p]),
    ('qgx_generator_info', C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_float)]),
    ('qgx_generator_wino_info', C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ('qgx_generator_wino_info_n', C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ('qgx_generator_layer2_kernel', C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ('qgx_generator_profile', C.c_int, [C.c_void_p, C.c_int]),
    ('qgx_generator_profile_read', C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ('qgx_noise_normal', C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_double, C.c_double, C.c_void_p]),
    ('qgx_last_error', C.c_char_p, []),
    ('qgx_version', C.c_char_p, []),
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: the HIP extension is not built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C '
            'pyqg_generative_amd/csrc`). There is no CPU fallback.')
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so) with the same soname as
    # /opt/rocm's, and this library links against that soname: whichever is loaded first serves both.
    # Load torch's first — torch finds no GPU when it is handed the other runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise QgxError(f'qgx error {rc}: {lib.qgx_last_error().decode()}')


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

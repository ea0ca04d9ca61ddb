"""ctypes binding of libgpe_hip.so (include/gpe_hip.h).  No torch types cross this boundary: device
buffers are passed as integer addresses (``tensor.data_ptr()``).

The library is REQUIRED: there is no CPU or eager fallback behind this module.  If the shared object is
missing or a symbol is absent the import of the engine fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

GPE_ABI_VERSION = 2
GPE_MAX_LAYERS = 12
GPE_MAX_ORTH = 4
GPE_MAX_DIM = 3
GPE_COMM_ID_BYTES = 128

GPE_OK = 0
GPE_ERR_INVALID, GPE_ERR_HIP, GPE_ERR_NONFINITE, GPE_ERR_STATE, GPE_ERR_NOMEM = -1, -2, -3, -4, -5

ACT_TANH, ACT_TANH_PLUS1 = 0, 1
POT_HARMONIC, POT_GAUSSIAN, POT_PERIODIC, POT_PRECOMPUTED, POT_NONE = 0, 1, 2, 3, 4
SCHED_CONST, SCHED_COSINE_LOSS, SCHED_PLATEAU = 0, 1, 2
PATH_AUTO, PATH_GENERIC, PATH_FUSED = 0, 1, 2
BASE_HERMITE, BASE_BOX, BASE_PRECOMPUTED = 0, 1, 2
ENV_NONE, ENV_SIN = 0, 1
RIESZ_PAPER, RIESZ_SUM, RIESZ_VARIATIONAL = 0, 1, 2
NET_MLP, NET_RESIDUAL = 0, 1
LAMBDA_RAYLEIGH, LAMBDA_ENERGY = 0, 1


class gpe_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_layers", C.c_int32), ("layers", C.c_int32 * GPE_MAX_LAYERS),
        ("activation", C.c_int32), ("complex_psi", C.c_int32), ("kinetic_coeff", C.c_float),
        ("potential", C.c_int32), ("pot_scale", C.c_float), ("omega", C.c_float * GPE_MAX_DIM),
        ("pot_a", C.c_float), ("pot_v0", C.c_float), ("pot_k", C.c_float), ("omega_rot", C.c_float),
        ("gamma", C.c_float), ("p", C.c_int32), ("abs_power", C.c_int32),
        ("base_mode", C.c_int32), ("base_deriv", C.c_int32), ("perturb_scale", C.c_float), ("bc_nn_scale", C.c_float),
        ("w_pde", C.c_float), ("w_bc", C.c_float), ("w_norm", C.c_float), ("w_sym", C.c_float), ("w_orth", C.c_float),
        ("sym_sign", C.c_float), ("dx", C.c_float), ("n_global", C.c_int64),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clip_norm", C.c_float),
        ("sched", C.c_int32), ("T_0", C.c_float), ("T_mult", C.c_float), ("eta_min", C.c_float),
        ("factor", C.c_float), ("patience", C.c_int32), ("min_lr", C.c_float), ("threshold", C.c_float),
        ("path", C.c_int32), ("world_size", C.c_int32), ("history_capacity", C.c_int32),
        ("stop_tol", C.c_float), ("stop_patience", C.c_int32),
        ("base_kind", C.c_int32), ("envelope", C.c_int32), ("box_L", C.c_float), ("env_L", C.c_float),
        ("w_riesz", C.c_float), ("riesz_kind", C.c_int32), ("net_kind", C.c_int32), ("lambda_kind", C.c_int32),
        ("w_reg_f", C.c_float), ("reg_f_eps", C.c_float), ("w_reg_lam", C.c_float), ("reg_lam_eps", C.c_float),
    ]


class gpe_scalars(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("loss", "pde", "bc", "norm", "sym", "orth", "mu", "num", "den", "sum_r2", "integral",
                 "grad_norm", "lr", "step", "nonfinite", "riesz", "reg")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_vp, _i64, _int, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float
_P = C.POINTER

# name -> (restype, argtypes).  Every symbol include/gpe_hip.h declares is listed here.
SYMBOLS = {
    "gpe_abi_version": (_int, []),
    "gpe_sizeof_config": (C.c_size_t, []),
    "gpe_sizeof_scalars": (C.c_size_t, []),
    "gpe_exchange_dbl_count": (_i64, []),
    "gpe_use_external_exchange": (_int, [_vp, _vp, _i64, _vp, _i64]),
    "gpe_create": (_int, [_P(gpe_config), _int, _vp, _P(_vp)]),
    "gpe_destroy": (None, [_vp]),
    "gpe_last_error": (C.c_char_p, [_vp]),
    "gpe_active_path": (_int, [_vp]),
    "gpe_active_kernels": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "gpe_comm_unique_id": (_int, [_vp, _vp]),
    "gpe_comm_init": (_int, [_vp, _vp, _int, _int]),
    "gpe_comm_destroy": (_int, [_vp]),
    "gpe_comm_info": (_int, [_vp, _P(_int), _P(_int), _P(_i64)]),
    "gpe_comm_set_async": (_int, [_vp, _int]),
    "gpe_step_dp": (_int, [_vp]),
    "gpe_run_dp": (_int, [_vp, _i64]),
    "gpe_param_count": (_i64, [_vp]),
    "gpe_set_params": (_int, [_vp, _vp, C.c_size_t]),
    "gpe_get_params": (_int, [_vp, _vp, C.c_size_t]),
    "gpe_get_grad": (_int, [_vp, _vp, C.c_size_t]),
    "gpe_get_adam_state": (_int, [_vp, _vp, _vp, C.c_size_t, _P(_i64)]),
    "gpe_set_adam_state": (_int, [_vp, _vp, _vp, C.c_size_t, _i64]),
    "gpe_reset_optimizer": (_int, [_vp, _f]),
    "gpe_bind_points": (_int, [_vp, _vp, _i64, _vp]),
    "gpe_bind_boundary": (_int, [_vp, _vp, _i64, _vp]),
    "gpe_bind_orth": (_int, [_vp, _int, _vp]),
    "gpe_bind_base": (_int, [_vp, _vp, _vp, _vp]),
    "gpe_forward": (_int, [_vp, _vp, _i64, _vp]),
    "gpe_forward_jets": (_int, [_vp, _vp, _i64, _vp]),
    "gpe_residual": (_int, [_vp, _P(gpe_scalars), _vp, _vp]),
    "gpe_eval_density": (_int, [_vp, _vp, _i64, _f, _int, _vp, _vp]),
    "gpe_step_begin": (_int, [_vp]),
    "gpe_step_backward": (_int, [_vp]),
    "gpe_step_update": (_int, [_vp]),
    "gpe_exchange_sums": (_int, [_vp, _P(_vp), _P(_i64)]),
    "gpe_exchange_grad": (_int, [_vp, _P(_vp), _P(_i64)]),
    "gpe_step": (_int, [_vp, _P(gpe_scalars)]),
    "gpe_run": (_int, [_vp, _i64]),
    "gpe_read_scalars": (_int, [_vp, _P(gpe_scalars)]),
    "gpe_read_history": (_int, [_vp, _i64, _i64, _P(gpe_scalars)]),
    "gpe_synchronize": (_int, [_vp]),
    "gpe_stop_state": (_int, [_vp, _P(_int), _P(_i64)]),
    "gpe_bind_target": (_int, [_vp, _vp]),
    "gpe_mse_begin": (_int, [_vp]),
    "gpe_mse_update": (_int, [_vp]),
    "gpe_mse_step": (_int, [_vp, _P(gpe_scalars)]),
    "gpe_mse_loss_grad": (_int, [_vp, _P(C.c_double)]),
    "gpe_set_gamma": (_int, [_vp, _f]),
    "gpe_set_power": (_int, [_vp, _int]),
    "gpe_set_lr": (_int, [_vp, _f]),
    "gpe_set_perturb_scale": (_int, [_vp, _f]),
    "gpe_set_n_global": (_int, [_vp, _i64]),
    "gpe_set_loss_weights": (_int, [_vp, _P(C.c_float)]),
    "gpe_profile_enable": (_int, [_vp, _int]),
    "gpe_profile_read": (_int, [_vp, _P(C.c_double)]),
    "gpe_step_cost": (_int, [_vp, _P(C.c_double), _P(C.c_double)]),
}

LIB_NAME = "libgpe_hip.so"


def library_path() -> str:
    # GPE_HIP_LIB: alternative build of the SAME library (kernel-tuning experiments); never a different backend
    return os.environ.get("GPE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


_lib = None


def load():
    """Load libgpe_hip.so and bind every symbol.  Raises (never falls back) when the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built.  Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.gpe_abi_version() != GPE_ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.gpe_abi_version()} != {GPE_ABI_VERSION}")
    if lib.gpe_sizeof_config() != C.sizeof(gpe_config) or lib.gpe_sizeof_scalars() != C.sizeof(gpe_scalars):
        raise ImportError(f"{path}: struct layout mismatch between include/gpe_hip.h and _capi.py")
    _lib = lib
    return lib

"""ctypes binding of libppsci_hip.so (C ABI: include/ppsci_hip.h).

The product path has NO CPU fallback: `lib()` raises if the gfx950 library is missing, and it
refuses a non-device build (`ppsci_is_device_build() == 0`, the CPU SIMT emulator that
tests/ uses to execute the kernel source in a GPU-less container) unless a test injected that
library explicitly through `_inject_for_tests`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

MAX_IN, MAX_DIRS, MAX_OUT, MAX_HIDDEN, MAX_PROG, MAX_RES, MAX_AUX = 8, 4, 8, 16, 128, 8, 16

ACT = {"tanh": 0, "silu": 1, "sin": 2, "sigmoid": 3, "cos": 4, "gelu": 5, "swish": 6, "stan": 7, "relu": 8, "leaky_relu": 9,
       "elu": 10, "selu": 11, "identity": 12}
PARAM_ACTS = ("swish", "stan")  # a trainable per-feature parameter vector per hidden layer (behind the last bias)
SIREN_W0 = 30.0  # activation.py:98
LINEAR_PLAIN, LINEAR_WEIGHT_NORM, LINEAR_RWF, LINEAR_FOURIER, LINEAR_BROADCAST = range(5)
LINEAR_PADDED = 100  # host-side record kind: ppsci_linear_pad / ppsci_linear_unpad (per-layer widths)
EMBED_NONE, EMBED_PERIOD, EMBED_STREAMS = 0, 1, 2

(OP_LD_IN, OP_LD_U, OP_LD_AUX, OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_POW, OP_SIN, OP_COS,
 OP_TANH, OP_EXP, OP_LOG, OP_SQRT, OP_ABS, OP_SINH, OP_COSH, OP_TAN, OP_MAX, OP_MIN, OP_SIGN, OP_HEAVISIDE,
 OP_DETACH, OP_ASIN, OP_ACOS, OP_ATAN, OP_ATAN2, OP_ASINH, OP_ACOSH, OP_ATANH, OP_ERF, OP_LGAMMA, OP_CEIL, OP_FLOOR,
 OP_LD_PARAM, OP_COUNT) = range(38)
MAX_EPARAM = 8
STEP_KEEP_FRAGMENTS = 1  # ppsci_taylor_step_run_ex flag (include/ppsci_hip.h)


class MlpDesc(C.Structure):
    _fields_ = [
        ("d_raw", C.c_int32), ("n_hidden", C.c_int32), ("width", C.c_int32), ("d_out", C.c_int32),
        ("activation", C.c_int32), ("skip_connection", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
        ("embed", C.c_int32 * MAX_IN), ("omega", C.c_float * MAX_IN), ("dirs", (C.c_float * MAX_IN) * MAX_DIRS),
        ("act_scale", C.c_float), ("fourier_half", C.c_int32), ("n3", C.c_int32), ("n4", C.c_int32),
    ]


class SpectralDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batch", "c_in", "c_out", "h", "wf", "modes_x", "modes_y")]


class PwVirtual(C.Structure):
    """ppsci_pw_virtual: an operand evaluated on load (mode 1: GELU(tensor); mode 2: GELU(W0 x0 + b0), never stored)."""
    _fields_ = [("mode", C.c_int32), ("K0", C.c_int32), ("x0", C.c_void_p), ("W0", C.c_void_p), ("b0", C.c_void_p)]


class ReduceSeg(C.Structure):
    """ppsci_reduce_seg: one row reduction of ppsci_reduce_rows_multi."""
    _fields_ = [("partials", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("accumulate", C.c_int32)]


class LinearJob(C.Structure):
    """ppsci_linear_job: one layer of ppsci_linear_multi."""
    _fields_ = [("kind", C.c_int32), ("fin", C.c_int32), ("fout", C.c_int32)] + [(n, C.c_void_p) for n in (
        "v", "g", "b", "W", "b_out", "gW", "gb", "gv", "gg", "gb_out")]


class ModMlpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_hidden", "width", "d_out", "activation")]


class SpinnGridDesc(C.Structure):
    _fields_ = [("n", C.c_int32 * 3), ("rank", C.c_int32), ("cu", C.c_float), ("cxx", C.c_float), ("cyy", C.c_float),
                ("czz", C.c_float), ("scale", C.c_float)]


class PirateEmbedDesc(C.Structure):
    _fields_ = [("d_raw", C.c_int32), ("d0", C.c_int32), ("half", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
                ("embed", C.c_int32 * MAX_IN), ("omega", C.c_float * MAX_IN), ("dirs", (C.c_float * MAX_IN) * MAX_DIRS),
                ("N", C.c_int64), ("NP", C.c_int64)]


PIRATE_ACT, PIRATE_GATE, PIRATE_RES = 0, 1, 2


class Instr(C.Structure):
    _fields_ = [("op", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_float)]


class Residual(C.Structure):
    _fields_ = [("value", C.c_int32), ("label", C.c_int32), ("weight", C.c_int32), ("area", C.c_int32),
                ("scale", C.c_float), ("kind", C.c_int32), ("scale_param", C.c_int32)]


class EpilogueDesc(C.Structure):
    _fields_ = [
        ("n_instr", C.c_int32), ("n_res", C.c_int32), ("n_streams", C.c_int32), ("n_in", C.c_int32),
        ("n_aux", C.c_int32), ("prog", Instr * MAX_PROG), ("res", Residual * MAX_RES),
    ]


class AdamArgs(C.Structure):  # ppsci_adam_args
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("grad_scale", C.c_float), ("step_t", C.c_int64)]


_HERE = os.path.dirname(os.path.abspath(__file__))
# PPSCI_HIP_LIB: another gfx950 build of the same sources (tools/build_variant.py: timer / ablation builds for measurements)
DEFAULT_LIB = os.environ.get("PPSCI_HIP_LIB") or os.path.join(_HERE, "libppsci_hip.so")
_lib: Optional[C.CDLL] = None
_injected = False

_SYMBOLS = {
    "ppsci_last_error": (C.c_char_p, []),
    "ppsci_is_device_build": (C.c_int, []),
    "ppsci_check_device": (C.c_int, []),
    "ppsci_linear_multi": (C.c_int, [C.c_int, C.POINTER(LinearJob), C.c_int, C.c_void_p]),
    "ppsci_reduce_rows_multi": (C.c_int, [C.c_int, C.POINTER(ReduceSeg), C.c_void_p]),
    "ppsci_reduce_rows_multi_adam": (C.c_int, [C.c_int, C.POINTER(ReduceSeg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_float,
                                               C.c_void_p]),
    "ppsci_release_fragments": (None, [C.c_void_p]),
    "ppsci_set_max_grid": (None, [C.c_int]),
    "ppsci_set_wide_min_nb": (None, [C.c_int]),
    "ppsci_set_bwd_accum": (None, [C.c_int]),
    "ppsci_set_bwd_layerwise": (None, [C.c_int]),
    "ppsci_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ppsci_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "ppsci_comm_world_size": (C.c_int, []),
    "ppsci_allreduce_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "ppsci_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ppsci_comm_destroy": (C.c_int, []),
    "ppsci_param_count": (C.c_int64, [C.POINTER(MlpDesc)]),
    "ppsci_stash_bytes": (C.c_int64, [C.POINTER(MlpDesc), C.c_int64]),
    "ppsci_bwd_partial_rows": (C.c_int64, [C.POINTER(MlpDesc), C.c_int64]),
    "ppsci_bwd_workspace_bytes": (C.c_int64, [C.POINTER(MlpDesc), C.c_int64]),
    "ppsci_epilogue_partial_rows": (C.c_int64, [C.c_int64]),
    "ppsci_taylor_fwd": (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "ppsci_epilogue": (C.c_int, [C.POINTER(EpilogueDesc), C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                 C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_epilogue_losses": (C.c_int, [C.POINTER(EpilogueDesc), C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                        C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_epilogue_params": (C.c_int, [C.POINTER(EpilogueDesc), C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                        C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "ppsci_taylor_bwd": (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_taylor_bwd_ws": (C.c_int, [C.POINTER(MlpDesc), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ppsci_dense_matvec": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "ppsci_reduce_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "ppsci_taylor_step_workspace_bytes": (C.c_int64, [C.POINTER(MlpDesc), C.POINTER(EpilogueDesc), C.c_int64]),
    "ppsci_taylor_step_kind": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(EpilogueDesc), C.c_int64]),
    "ppsci_set_fused_step": (None, [C.c_int]),
    "ppsci_set_step_tail": (None, [C.c_int]),
    "ppsci_set_fast_program": (None, [C.c_int]),
    "ppsci_taylor_step_run_ex": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(AdamArgs), C.c_void_p, C.c_int]),
    "ppsci_taylor_step_plan_apply": (C.c_int, [C.c_void_p, C.POINTER(AdamArgs), C.c_void_p]),
    "ppsci_set_static_program": (None, [C.c_int]),
    "ppsci_epilogue_predecode": (C.c_int, [C.POINTER(EpilogueDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "ppsci_taylor_step_plan_static": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p)]),
    "ppsci_taylor_step_plan": (C.c_void_p, [C.POINTER(MlpDesc), C.POINTER(EpilogueDesc), C.c_void_p, C.c_int64,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ppsci_taylor_step_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(AdamArgs), C.c_void_p]),
    "ppsci_taylor_step_plan_set_scales": (C.c_int, [C.c_void_p, C.POINTER(EpilogueDesc)]),
    "ppsci_taylor_step_run_main": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppsci_taylor_step_plan_free": (None, [C.c_void_p]),
    "ppsci_taylor_step": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(EpilogueDesc), C.c_void_p, C.c_int64, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(AdamArgs), C.c_void_p]),
    "ppsci_spectral_conv2d_fwd": (C.c_int, [C.POINTER(SpectralDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "ppsci_spectral_conv2d_bwd": (C.c_int, [C.POINTER(SpectralDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_spectral_conv2d_bwd_real": (C.c_int, [C.POINTER(SpectralDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "ppsci_spectral_conv2d_fwd_scaled": (C.c_int, [C.POINTER(SpectralDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_float, C.c_int, C.c_void_p]),
    "ppsci_spectral_conv2d_bwd_real_scaled": (C.c_int, [C.POINTER(SpectralDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                                        C.c_float, C.c_int, C.c_void_p]),
    "ppsci_dft2_kept_supported": (C.c_int, [C.c_int] * 4),
    "ppsci_dft2_kept_fwd": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 3),
    "ppsci_dft2_kept_inv": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 3),
    "ppsci_dft2_kept_inv_stats": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]),
    "ppsci_fno_tail_fwd_ex": (C.c_int, [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 9 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]),
    "ppsci_fno_tail_bwd_ex": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 13 + [C.c_int] * 4 + [C.c_void_p] * 3),
    "ppsci_spectral_conv2d_fwd_kept": (C.c_int, [C.POINTER(SpectralDesc)] + [C.c_void_p] * 4 + [C.c_float, C.c_void_p]),
    "ppsci_spectral_conv2d_inv_kept_ex": (C.c_int, [C.POINTER(SpectralDesc)] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_spectral_conv2d_inv_kept": (C.c_int, [C.POINTER(SpectralDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_spectral_conv2d_bwd_kept": (C.c_int, [C.POINTER(SpectralDesc)] + [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_float,
                                                                                                 C.c_void_p]),
    "ppsci_fft2d_r2c": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_fft2d_c2r": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_pw_conv": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_pw_conv_v": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(PwVirtual), C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.POINTER(PwVirtual), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_fno_proj_hidden_grad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_fno_lift0_wgrad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "ppsci_pw_conv_wgrad_v": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(PwVirtual), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ppsci_set_pw_pixels_per_lane": (None, [C.c_int]),
    "ppsci_pw_conv_wgrad_chunks": (C.c_int64, [C.c_int, C.c_int]),
    "ppsci_fno_lift0_wgrad_chunks": (C.c_int64, [C.c_int, C.c_int]),
    "ppsci_pw_conv_wgrad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "ppsci_pad2d": (C.c_int, [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_dft2_kept_from_supported": (C.c_int, [C.c_int] * 4),
    "ppsci_dft2_kept_fwd_from": (C.c_int, [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_dft2_kept_inv_from": (C.c_int, [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_sht_supported": (C.c_int, [C.c_int] * 4),
    "ppsci_sht_analysis": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 5),
    "ppsci_sht_synthesis": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 5),
    "ppsci_sht_synthesis_contract": (C.c_int, [C.c_int] * 8 + [C.c_void_p] * 7),
    "ppsci_sht_contract": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p]),
    "ppsci_sht_contract_wgrad": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 5),
    "ppsci_spectrum_resize": (C.c_int, [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_resample2d_supported": (C.c_int, [C.c_int] * 4),
    "ppsci_resample2d": (C.c_int, [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]),
    "ppsci_field_loss_sums": (C.c_int, [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "ppsci_field_loss_finish": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float] + [C.c_void_p] * 4),
    "ppsci_field_loss_adjoint": (C.c_int, [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "ppsci_tanh_fwd": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_tanh_bwd": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ppsci_fno_tail_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "ppsci_fno_tail_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_modmlp_param_count": (C.c_int64, [C.POINTER(ModMlpDesc)]),
    "ppsci_modmlp_bwd_rows": (C.c_int64, [C.POINTER(ModMlpDesc), C.c_int64]),
    "ppsci_modmlp_bwd_parts_supported": (C.c_int, [C.POINTER(ModMlpDesc), C.POINTER(SpinnGridDesc)]),
    "ppsci_modmlp_bwd_batch_parts": (C.c_int, [C.POINTER(ModMlpDesc), C.POINTER(SpinnGridDesc), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.c_int64, C.c_void_p]),
    "ppsci_set_modmlp_tile": (None, [C.c_int]),
    "ppsci_modmlp_stash_floats": (C.c_int64, [C.POINTER(ModMlpDesc), C.c_int64]),
    "ppsci_modmlp_fwd": (C.c_int, [C.POINTER(ModMlpDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "ppsci_modmlp_bwd": (C.c_int, [C.POINTER(ModMlpDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "ppsci_modmlp_fwd_batch": (C.c_int, [C.POINTER(ModMlpDesc), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "ppsci_modmlp_bwd_batch": (C.c_int, [C.POINTER(ModMlpDesc), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.c_int64, C.c_void_p]),
    "ppsci_pirate_embed_fwd": (C.c_int, [C.POINTER(PirateEmbedDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_pirate_embed_chunks": (C.c_int64, [C.c_int64]),
    "ppsci_pirate_embed_bwd": (C.c_int, [C.POINTER(PirateEmbedDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "ppsci_pirate_act_chunks": (C.c_int64, [C.c_int64]),
    "ppsci_pirate_act_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "ppsci_pirate_act_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 14),
    "ppsci_pirate_out_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_pirate_out_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_spinn_grid_partial_rows": (C.c_int64, [C.POINTER(SpinnGridDesc)]),
    "ppsci_spinn_grid_fwd": (C.c_int, [C.POINTER(SpinnGridDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_spinn_grid_bwd_scratch_floats": (C.c_int64, [C.POINTER(SpinnGridDesc)]),
    "ppsci_spinn_grid_bwd": (C.c_int, [C.POINTER(SpinnGridDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_adam_step": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_int64, C.c_float, C.c_void_p]),
    "ppsci_optim_step": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "ppsci_causal_weights": (C.c_int, [C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppsci_linear_materialize": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "ppsci_linear_pad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "ppsci_linear_unpad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "ppsci_linear_pullback": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SYMBOLS)


def _bind(path: str) -> C.CDLL:
    lib_ = C.CDLL(path)
    for name, (res, args) in _SYMBOLS.items():
        fn = getattr(lib_, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib_


def lib() -> C.CDLL:
    """The loaded gfx950 library.  Raises loudly when it is missing -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(DEFAULT_LIB):
            raise RuntimeError(
                f"{DEFAULT_LIB} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). paddlescience_amd has no CPU fallback."
            )
        l = _bind(DEFAULT_LIB)
        if l.ppsci_is_device_build() != 1:
            raise RuntimeError(f"{DEFAULT_LIB} is not a gfx950 device build")
        import torch

        if torch.cuda.is_available() and l.ppsci_check_device() != 0:
            raise RuntimeError(l.ppsci_last_error().decode())
        _lib = l
    return _lib


def _inject_for_tests(path: Optional[str]) -> None:
    """tests/ only: run the kernel source under the CPU SIMT emulator build."""
    global _lib, _injected
    _lib = _bind(path) if path else None
    _injected = path is not None


def is_emulated() -> bool:
    return _injected


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().ppsci_last_error()
        raise RuntimeError(f"libppsci_hip error {rc}: {msg.decode() if msg else ''}")


def ptr_array(ptrs: Sequence[int]):
    arr = (C.c_void_p * max(1, len(ptrs)))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def make_mlp_desc(d_raw: int, n_hidden: int, width: int, d_out: int, activation: str, skip_connection: bool,
                  dirs: Sequence[Sequence[float]], n2: int, embed: Optional[Sequence[int]] = None,
                  omega: Optional[Sequence[float]] = None, fourier_half: int = 0, n3: int = 0, n4: int = 0) -> MlpDesc:
    d = MlpDesc()
    d.n3, d.n4 = int(n3), int(n4)
    d.fourier_half = int(fourier_half)
    d.d_raw, d.n_hidden, d.width, d.d_out = d_raw, n_hidden, width, d_out
    if activation == "siren":  # sin(30 z): the sin kernels with the pre-activation multiplier
        activation, d.act_scale = "sin", SIREN_W0
    if activation not in ACT:
        raise NotImplementedError(f"activation {activation!r} has no HIP kernel (supported: {sorted(ACT)})")
    d.activation = ACT[activation]
    d.skip_connection = 1 if skip_connection else 0
    d.n1, d.n2 = len(dirs), n2
    if len(dirs) > MAX_DIRS:
        raise NotImplementedError(f"at most {MAX_DIRS} derivative directions per launch, got {len(dirs)}")
    for i, row in enumerate(dirs):
        for j, v in enumerate(row):
            d.dirs[i][j] = float(v)
    for j in range(d_raw):
        d.embed[j] = int(embed[j]) if embed is not None else 0
        d.omega[j] = float(omega[j]) if omega is not None else 0.0
    return d

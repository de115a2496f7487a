"""ctypes binding of libmnrf_hip.so (the C ABI declared in include/mnrf.h).

There is no CPU fallback: if the shared library is missing, or a call fails,
a RuntimeError is raised.  torch is imported first so that the HIP runtime
already loaded by torch (same SONAME libamdhip64.so.7) is the one the library
binds to -- device pointers and streams then belong to one runtime.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MNRF_LIB", os.path.join(_HERE, "libmnrf_hip.so"))   # MNRF_LIB: tuning experiments
CSRC = os.path.join(_HERE, "csrc")

_c_f = ctypes.c_void_p      # device float*
_c_i = ctypes.c_void_p      # device int32*
_i64 = ctypes.c_int64
_int = ctypes.c_int
_u32 = ctypes.c_uint
_flt = ctypes.c_float
_str = ctypes.c_void_p      # hipStream_t

MNRF_SIGMA_ONLY = 1
MNRF_GRAD_NORMAL = 2
MNRF_SPLIT_F16 = 4
MNRF_TCNN_VALU = 8
MNRF_TCNN_F16 = 256
MNRF_TCNN_GRAD_FIXED = 512
MNRF_TCNN_TABLE_F16 = 1024
MNRF_CUT_NORMAL_HEAD = 32
MNRF_CUT_MIRROR_HEAD = 64
MNRF_DW_ACCUMULATE = 128
MNRF_TRAIN_PLANES = 256
MNRF_PLANES_Y_HALF = 0x100000
MNRF_TCNN_GRAD_F16 = 16
MNRF_DETACH_W_MASK = 1
MNRF_DETACH_W_NORMAL = 2
N_PARAMS = 32

# name -> (restype, argtypes): exactly the prototypes of include/mnrf.h
SIGNATURES = {
    "mnrf_last_error": (ctypes.c_char_p, []),
    "mnrf_version": (_int, []),
    "mnrf_packed_floats": (_i64, []),
    "mnrf_pack_weights": (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _str]),
    "mnrf_pack_weights_n": (_int, [_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _str]),
    "mnrf_embed": (_int, [_c_f, _i64, _int, _int, _c_f, _str]),
    "mnrf_field_forward": (_int, [_c_f, _u32, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _i64,
                                  _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _str]),
    "mnrf_sample_coarse": (_int, [_c_f, _i64, _c_f, _int, _int, _flt, _c_f, _c_f, _str]),
    "mnrf_composite": (_int, [_c_f, _i64, _int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _int,
                              _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _str]),
    "mnrf_composite_backward": (_int, [_c_f, _i64, _int] + [_c_f] * 7 + [_int] + [_c_f] * 17 + [_int, _c_f, _str]),
    "mnrf_reflect_backward": (_int, [_c_f, _c_f, _c_i, _i64, _c_f, _i64, _c_f, _c_f, _c_f, _str]),
    "mnrf_blend_backward": (_int, [_c_f, _c_f, _c_i, _i64, _i64, _int, _c_f, _c_f, _str]),
    "mnrf_embed_backward": (_int, [_c_f, _c_f, _i64, _int, _int, _c_f, _str]),
    "mnrf_ray_grads": (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f, _str]),
    "mnrf_train_save_floats": (_i64, [_i64]),
    "mnrf_train_mask_words": (_i64, [_i64]),
    "mnrf_train_workspace_floats": (_i64, [_i64]),
    "mnrf_field_forward_train": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _i64] + [_c_f] * 9 + [_u32, _str]),
    "mnrf_train_workspace2_floats": (_i64, [_i64]),
    "mnrf_field_backward2": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _c_f, _c_f, _c_f, _c_f,
                                    ctypes.POINTER(ctypes.c_void_p), _c_f, _u32, _str]),
    "mnrf_field_backward": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int] + [_c_f] * 11 +
                            [ctypes.POINTER(ctypes.c_void_p), _c_f, _c_f, _c_f, _u32, _str]),
    "mnrf_fused_samples_per_ray": (_int, []),
    "mnrf_field_composite_fused": (_int, [_c_f, _i64, _c_f, _c_f, _c_f, _i64, _int] + [_c_f] * 7 + [_str]),
    "mnrf_train_planes_bytes": (_i64, [_i64]),
    "mnrf_train_dy_planes_bytes": (_i64, [_i64]),
    "mnrf_field_backward_planes": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int] + [_c_f] * 9 +
                                   [ctypes.c_void_p, ctypes.c_void_p, _c_f, _c_f, _c_f, _u32, _str]),
    "mnrf_dw_planes_workspace_floats": (_i64, [_int, ctypes.POINTER(ctypes.c_int64)]),
    "mnrf_dw_planes": (_int, [_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                              ctypes.POINTER(ctypes.c_void_p), _c_f, ctypes.POINTER(ctypes.c_void_p), _int, _str]),
    "mnrf_train_planes2_bytes": (_i64, [_i64]),
    "mnrf_train_dy_planes2_bytes": (_i64, [_i64]),
    "mnrf_field_backward2_planes": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _c_f, _c_f, _c_f,
                                           ctypes.c_void_p, ctypes.c_void_p, _c_f, _c_f, _str]),
    "mnrf_dw_planes2_workspace_floats": (_i64, [_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]),
    "mnrf_dw_planes2": (_int, [_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), _c_f, ctypes.POINTER(ctypes.c_void_p),
                               _int, _str]),
    "mnrf_adam_step": (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _flt, ctypes.c_double, ctypes.c_double, _flt, _flt, _i64, _c_i, _c_f, _c_f, _str]),
    "mnrf_bench_stream": (_int, [ctypes.c_void_p, _i64, _int, _int, _str]),
    "mnrf_bench_stream2": (_int, [ctypes.c_void_p, ctypes.c_void_p, _int, _i64, _i64, _int, _int, _int, _int, _str]),
    "mnrf_bench_gather": (_int, [ctypes.c_void_p, _i64, _int, _i64, _int, _c_f, _str]),
    "mnrf_tcnn_encode": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _i64, _c_f, _i64, _c_f, _c_f, _int,
                                _c_f, _str]),
    "mnrf_tcnn_encode_flags": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _i64, _c_f, _i64, _c_f, _c_f, _int,
                                      _c_f, _u32, _str]),
    "mnrf_tcnn_table_half": (_int, [_c_f, _i64, ctypes.c_void_p, _str]),
    "mnrf_tcnn_weight_floats": (_int, []),
    "mnrf_tcnn_pack_weights": (_int, [ctypes.POINTER(ctypes.c_void_p), _c_f, _str]),
    "mnrf_tcnn_forward": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _c_f, _u32, _i64, _c_f, _i64,
                                 _c_f, _c_f, _int, _c_f, _i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _str]),
    "mnrf_tcnn_backward": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _c_f, _i64, _c_f, _i64,
                                  _c_f, _c_f, _int, _c_f, _i64] + [_c_f] * 11 + [_u32, _str]),
    "mnrf_tcnn_forward_n": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _c_f, _u32, _i64, _c_f, _i64,
                                   _c_f, _c_f, _int, _c_f, _i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_tcnn_backward_n": (_int, [_c_f, ctypes.POINTER(ctypes.c_int64), ctypes.c_double, _int, _flt, _c_f, _i64, _c_f, _i64,
                                    _c_f, _c_f, _int, _c_f, _i64] + [_c_f] * 11 + [_u32, _c_i, _str]),
    "mnrf_tcnn_backward_workspace_floats": (_i64, [ctypes.POINTER(ctypes.c_int64)]),
    "mnrf_tcnn_backward_workspace_floats2": (_i64, [ctypes.POINTER(ctypes.c_int64), ctypes.c_uint]),
    "mnrf_tcnn_backward_workspace_floats3": (_i64, [ctypes.POINTER(ctypes.c_int64), ctypes.c_uint, _i64]),
    "mnrf_sample_fine": (_int, [_c_f, _c_f, _i64, _int, _c_f, _int, _int, _c_f, _str]),
    "mnrf_threshold_mask": (_int, [_c_f, _i64, _c_i, _str]),
    "mnrf_reflect_compact": (_int, [_c_f, _c_f, _c_f, _c_f, _flt, _c_f, _i64, _int, _flt, _c_f, _c_i,
                                    _c_i, _c_f, _str]),
    "mnrf_blend_scatter": (_int, [_c_f, _c_f, _c_i, _i64, _c_f, _i64, _int, _c_f, _c_f, _str]),
    "mnrf_generate_rays": (_int, [_int, _int, _flt, ctypes.POINTER(ctypes.c_float), _flt, _flt, _c_f, _str]),
    "mnrf_loss_workspace_floats": (_i64, [_i64, _int, _int, _i64]),
    "mnrf_total_loss": (_int, [ctypes.c_void_p, _c_f, _str]),     # const MnrfLossArgs* (losses._Args)
    "mnrf_loss_count": (_int, [ctypes.c_void_p, _c_f, _str]),
    "mnrf_mse_blocks": (_int, []),
    "mnrf_mse_psnr": (_int, [_c_f, _c_f, ctypes.c_void_p, _i64, _int, _c_f, _c_f, _str]),
    # ---- live row counts on the device (round 5): the namesake's arguments + n_live (device int32) in front of the stream
    "mnrf_embed_n": (_int, [_c_f, _i64, _int, _int, _c_f, _c_i, _str]),
    "mnrf_embed_backward_n": (_int, [_c_f, _c_f, _i64, _int, _int, _c_f, _c_i, _str]),
    "mnrf_sample_coarse_n": (_int, [_c_f, _i64, _c_f, _int, _int, _flt, _c_f, _c_f, _c_i, _str]),
    "mnrf_ray_prologue_n": (_int, [_c_f, _i64, _int, _c_f, _int, _int, _flt, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_ray_fan_backward_n": (_int, [_c_f] * 7 + [_i64, _int, _c_f, _c_i, _str]),
    "mnrf_composite_n": (_int, [_c_f, _i64, _int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _int,
                                _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_composite_sample_n": (_int, [_c_f, _i64, _int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _int,
                                       _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _int, _int, _c_f, _c_i, _str]),
    "mnrf_composite_backward_n": (_int, [_c_f, _i64, _int] + [_c_f] * 7 + [_int] + [_c_f] * 17 + [_int, _c_f, _c_i, _str]),
    "mnrf_sample_fine_n": (_int, [_c_f, _c_f, _i64, _int, _c_f, _int, _int, _c_f, _c_i, _str]),
    "mnrf_threshold_mask_n": (_int, [_c_f, _i64, _c_i, _c_i, _str]),
    "mnrf_reflect_compact_n": (_int, [_c_f, _c_f, _c_f, _c_f, _flt, _c_f, _i64, _int, _flt, _c_f, _c_i, _c_i, _c_f, _c_i, _c_i, _str]),
    "mnrf_blend2_n": (_int, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _i64, _int, _c_f, _c_f, _c_i, _str]),
    "mnrf_blend2_backward_n": (_int, [_c_f, _c_f, _c_i, _c_f, _i64, _int, _c_f, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_blend_scatter_n": (_int, [_c_f, _c_f, _c_i, _i64, _c_f, _i64, _int, _c_f, _c_f, _c_i, _c_i, _str]),
    "mnrf_reflect_backward_gather_n": (_int, [_c_f, _c_f, _c_i, _c_f, _i64, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_reflect_backward_n": (_int, [_c_f, _c_f, _c_i, _i64, _c_f, _i64, _c_f, _c_f, _c_f, _c_i, _str]),
    "mnrf_blend_backward_n": (_int, [_c_f, _c_f, _c_i, _i64, _i64, _int, _c_f, _c_f, _c_i, _c_i, _str]),
    "mnrf_ray_grads_n": (_int, [_c_f, _c_f, _c_f, _i64, _int, _c_f, _c_f, _c_i, _str]),
    "mnrf_field_forward_train_n": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _i64] + [_c_f] * 9 + [_u32, _c_i, _str]),
    "mnrf_field_backward_planes_n": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int] + [_c_f] * 9 +
                                     [ctypes.c_void_p, ctypes.c_void_p, _c_f, _c_f, _c_f, _u32, _c_i, _str]),
    "mnrf_field_backward2_planes_n": (_int, [_c_f, _i64, _c_f, _i64, _c_f, _c_f, _int, _c_f, _c_f, _c_f, _c_f,
                                             ctypes.c_void_p, ctypes.c_void_p, _c_f, _c_f, _c_i, _str]),
    "mnrf_dw_planes2_n_workspace_floats": (_i64, [_int]),
    "mnrf_dw_planes2_n": (_int, [_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.POINTER(ctypes.c_int), _c_f, ctypes.POINTER(ctypes.c_void_p), _int, _str]),
    "mnrf_adam_prep": (_int, [ctypes.c_void_p, ctypes.c_void_p, _c_i, _c_f, _c_f, ctypes.POINTER(ctypes.c_void_p), _int, _c_f, _str]),
    "mnrf_adam_step_dev": (_int, [_c_f, _c_f, _c_f, _c_f, _i64, _c_f, _c_i, _str]),
    "mnrf_adam_step_dev_n": (_int, [_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), _c_f, ctypes.POINTER(ctypes.c_void_p), _str]),
}

_lib = None


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libmnrf_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-j4", "-C", CSRC], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("building libmnrf_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library with typed entry points; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension must be built "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().mnrf_last_error()
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be fp32/int32, contiguous, on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mirror_nerf_amd runs on the GPU only (tensor is on %s)" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    if t.device.index != _cur_device():
        # kernels are launched on the CURRENT device's current stream (stream()): one process per GPU binds it once with
        # torch.cuda.set_device(LOCAL_RANK) (dist.init_from_env); anything else must wrap calls in torch.cuda.device(...)
        raise RuntimeError(f"tensor lives on cuda:{t.device.index} but the current device is cuda:{_cur_device()}; "
                           "call torch.cuda.set_device / use `with torch.cuda.device(t.device)`")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def stream():
    """The current torch stream of the current device as a hipStream_t (every kernel is launched on it)."""
    if _raw_stream is not None:      # ~1 us; torch.cuda.current_stream() builds a Stream object (~13 us, 20 launches per step)
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

"""ctypes binding of include/audiopure.h (audiopure_amd/lib/libaudiopure_hip.so).

There is deliberately NO fallback: if the library is missing or a call fails the
caller gets an exception, never a silently different code path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AUDIOPURE_HIP_LIB: load another build of the same sources (tools/*.py point it at the -DAP_TOOLS library, which adds
# the timing-only ap_debug_* hooks); every symbol of include/audiopure.h is still required from it.
LIB_PATH = os.environ.get("AUDIOPURE_HIP_LIB") or os.path.join(_HERE, "lib", "libaudiopure_hip.so")

AP_PREC_F32 = 0
AP_PREC_BF16 = 1
AP_PREC_F32_SPLIT = 2
AP_PREC_BF16_STORE = 4


class NativeError(RuntimeError):
    pass


class ApConfig(C.Structure):
    _fields_ = [("res_channels", C.c_int32), ("skip_channels", C.c_int32), ("num_res_layers", C.c_int32),
                ("dilation_cycle", C.c_int32), ("embed_dim_in", C.c_int32), ("embed_dim_mid", C.c_int32),
                ("embed_dim_out", C.c_int32), ("T", C.c_int32), ("beta_0", C.c_float), ("beta_T", C.c_float),
                ("precision", C.c_int32)]


class ApStep(C.Structure):
    _fields_ = [("step", C.c_float), ("ca", C.c_float), ("cb", C.c_float), ("cs", C.c_float), ("draw", C.c_int32)]


_vp, _fp, _u64, _u32, _i, _f, _sz = C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/audiopure.h declares
SIGNATURES = {
    "ap_last_error": (C.c_char_p, []),
    "ap_version": (_i, []),
    "ap_ctx_create": (_i, [C.POINTER(ApConfig), C.POINTER(_vp)]),
    "ap_ctx_destroy": (_i, [_vp]),
    "ap_ctx_set_schedule": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), _i]),
    "ap_ctx_set_sde_schedule": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), _i]),
    "ap_wavenet_blob_elems": (_sz, [C.POINTER(ApConfig)]),
    "ap_ctx_load_wavenet": (_i, [_vp, _fp, _sz, _fp, _vp]),
    "ap_ctx_get_folded": (_i, [_vp, _i, _i, _fp, _sz, _vp]),
    "ap_ctx_get_schedule": (_i, [_vp, _i, C.POINTER(_f), _i]),
    "ap_profile_enable": (_i, [_vp, _i]),
    "ap_profile_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ap_conv2d_set_workspace": (_i, [_vp, _sz]),
    "ap_conv_profile_enable": (_i, [_i]),
    "ap_conv_profile_read": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _i]),
    "ap_conv_profile_launch": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ap_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ap_embed": (_i, [_vp, _f, _fp, _vp]),
    "ap_init_conv": (_i, [_vp, _fp, _fp, _i, _i, _vp]),
    "ap_resblock_fwd": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_resblock_fwd_save": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_resblock_bwd": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "ap_resblock_bwd_available": (_i, [_vp, _i, _i]),
    "ap_ctx_prepare_backward": (_i, [_vp, _vp]),
    "ap_init_conv_u": (_i, [_vp, _fp, _fp, _vp, _i, _i, _vp]),
    "ap_resblock_fwd_u": (_i, [_vp, _i, _vp, _fp, _vp, _vp, _i, _i, _vp]),
    "ap_resblock_fwd_u_save": (_i, [_vp, _i, _vp, _fp, _vp, _vp, _vp, _i, _i, _vp]),
    "ap_resblock_bwd_bf16": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _vp, _fp, _i, _i, _vp]),
    "ap_resblock_bwd_bf16_available": (_i, [_vp, _i, _i]),
    "ap_gate_factor_bytes": (_sz, [_i, _i]),
    "ap_resblock_fwd_gate_save": (_i, [_vp, _i, _fp, _fp, _fp, _vp, _vp, _i, _i, _vp]),
    "ap_resblock_bwd_bf16_saved": (_i, [_vp, _i, _vp, _fp, _vp, _i, _vp, _fp, _i, _i, _vp]),
    "ap_bwd_bf16_rows_image": (_i, [_fp, _vp, _i, _i, _i, _vp]),
    "ap_resblock_fwd_gate": (_i, [_vp, _i, _fp, _fp, _fp, _vp, _i, _i, _vp]),
    "ap_skip_gemm": (_i, [_vp, _i, _i, _vp, _fp, _i, _i, _i, _vp]),
    "ap_ctx_set_skip_group": (_i, [_vp, _i]),
    "ap_ctx_set_f32_form": (_i, [_vp, _i]),
    "ap_ctx_get_f32_form": (_i, [_vp]),
    "ap_profile_is_enabled": (_i, [_vp]),
    "ap_profile_read_split": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ap_final_affine": (_i, [_vp, _fp, _fp, _fp, _fp, _f, _f, _f, _fp, _u64, _u32, _u64, _i, _i, _vp]),
    "ap_affine_noise": (_i, [_fp, _fp, _f, _f, _fp, _u64, _u32, _u64, _i, _i, _vp]),
    "ap_eps_fwd": (_i, [_vp, _fp, _f, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_eps_affine": (_i, [_vp, _fp, _f, _f, _f, _fp, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_purify_chain": (_i, [_vp, _fp, _f, _f, C.POINTER(ApStep), _i, _fp, _u64, _u64, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_purify_ddpm": (_i, [_vp, _fp, _i, _i, _fp, _u64, _u64, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_purify_sde": (_i, [_vp, _fp, _i, _fp, _u64, _u64, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_one_shot_denoise": (_i, [_vp, _fp, _i, _fp, _i, _i, _vp, _sz, _vp]),
    "ap_m5_create": (_i, [_i, _i, _i, _i, _f, _fp, _sz, _vp, C.POINTER(_vp)]),
    "ap_m5_destroy": (_i, [_vp]),
    "ap_m5_blob_elems": (_sz, [_i, _i, _i]),
    "ap_m5_fwd": (_i, [_vp, _fp, _fp, _i, _i, _vp]),
    "ap_melspec_db": (_i, [_fp, _fp, _i, _i, _i, _i, _vp]),
    "ap_kws_blob_elems": (C.c_size_t, [_i, _i, _i]),
    "ap_kws_create": (_i, [_i, _i, _i, _fp, _sz, _vp, C.POINTER(C.c_void_p)]),
    "ap_kws_destroy": (_i, [_vp]),
    "ap_kws_fwd": (_i, [_vp, _fp, _fp, _i, _i, _vp]),
    "ap_melspec_db_htk": (_i, [_fp, _fp, _i, _i, _i, _vp]),
    "ap_kws_bwd_scratch_elems": (C.c_size_t, [_vp, _i, _i]),
    "ap_kws_bwd": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "ap_melspec_db_htk_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_nes_perturb": (_i, [_fp, _fp, _f, _u64, _u32, _i, _i, _i, _i, _vp]),
    "ap_nes_grad": (_i, [_fp, _fp, _u64, _u32, _i, _i, _i, _i, _vp]),
    "ap_argmax_hist": (_i, [_fp, _vp, _i, _i, _vp]),
    "ap_acc_channels": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_relu_mask": (_i, [_fp, _fp, _fp, _sz, _vp]),
    "ap_zero_insert2d": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_pool2d_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_melspec_db_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_m5_bwd": (_i, [_vp, _fp, _fp, _fp, _i, _i, _vp]),
    "ap_gate_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_relu_outer_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_init_conv_bwd": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "ap_conv2d_packed_elems": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "ap_conv2d_pack": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "ap_conv2d_fwd": (_i, [_fp, _fp, _fp, _fp, _fp] + [_i] * 13 + [_vp]),
    "ap_conv2d_fwd_slice": (_i, [_fp, _fp, _fp, _fp, _fp] + [_i] * 15 + [_vp]),
    "ap_affine_nchw": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_add_nchw": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_copy_channels": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_pool2d": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ap_groupnorm_nchw": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _i, _vp]),
    "ap_timestep_embedding": (_i, [_fp, _fp, _fp, _i, _i, _vp]),
    "ap_silu": (_i, [_fp, _fp, _sz, _vp]),
    "ap_upsample_nearest2x": (_i, [_fp, _fp, _i, _i, _i, _vp]),
    "ap_attention_qkv": (_i, [_fp, _fp, _i, _i, _i, _i, _vp]),
    "ap_groupnorm_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _i, _vp]),
    "ap_attention_qkv_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "ap_upsample_nearest2x_bwd": (_i, [_fp, _fp, _i, _i, _i, _vp]),
    "ap_axpbyc": (_i, [_fp, _fp, _fp, _f, _f, _f, _sz, _vp]),
    "ap_psample_update": (_i, [_fp, _fp, _fp, _fp, _f, _f, _f, _f, _f, _i, _sz, _vp]),
    "ap_philox_normal": (_i, [_fp, _u64, _u32, _u64, _i, _i, _vp]),
}

_LIB = None


def lib():
    """Load the HIP library (once) and bind every declared symbol; raise if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). audiopure_amd has no CPU fallback.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = l
    return l


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ap_last_error()
        raise NativeError(f"{what} failed with code {rc}: {msg.decode(errors='replace') if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a contiguous fp32 CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    import torch
    assert isinstance(t, torch.Tensor)
    if not t.is_cuda:
        raise NativeError("audiopure_amd ops need device (cuda/HIP) tensors; got a CPU tensor and there is no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise NativeError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    if t.device.index != torch.cuda.current_device():
        # kernels, hipMalloc and stream() bind to the CURRENT device; a pointer from another one would fault or,
        # with peer access, silently run cross-device with no stream ordering
        raise NativeError(f"tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                          "enter `with torch.cuda.device(t.device):` (the module entry points do: _native.on_device)")
    return t.data_ptr()


def on_device(fn):
    """Decorator for module entry points: run the call with the HIP device of the first CUDA tensor argument (or of the
    module's parameters) current, so that streams, allocations and launches of the library all bind to that device."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        import torch
        dev = None
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dev = a.device
                break
        if dev is None and isinstance(self, torch.nn.Module):
            p = next(self.parameters(), None)
            if p is not None and p.is_cuda:
                dev = p.device
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kw)
        with torch.cuda.device(dev):
            return fn(self, *args, **kw)
    return wrapper


_CONV_WS = {}                                    # device index -> the tensor the library's split-K path writes its partial sums to
CONV_WS_BYTES = 128 << 20                        # (the K slices of the UNet's 8 x 8 maps at B = 256: 4 x 16.8 MB)


def use_conv_workspace(device) -> None:
    """Register this device's split-K workspace with the library (``ap_conv2d_set_workspace``: one slot per HIP device, keyed by
    the device current at the call and at each launch; allocated once per device by torch's caching allocator and kept alive
    here -- the library itself allocates nothing).  Called at the entry of every native conv-net forward and of the 1-D
    backward GEMMs; allocates once, registers every time."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        ws = _CONV_WS.get(idx)
        if ws is None:
            ws = _CONV_WS[idx] = torch.empty(CONV_WS_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", idx))
        # registered on EVERY call (one cheap ctypes call): the library's per-device slot can be cleared behind this cache
        # (ap_conv2d_set_workspace(NULL, 0)), and split-K must not then stay off for the rest of the process
        check(lib().ap_conv2d_set_workspace(ws.data_ptr(), CONV_WS_BYTES), "ap_conv2d_set_workspace")


def stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def farr(values):
    a = (C.c_float * len(values))(*[float(v) for v in values])
    return a

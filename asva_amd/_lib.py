"""ctypes binding of libavsd_hip.so (declared in include/avsd.h).

The library is the only compute path: there is no fallback.  Importing this module is cheap; the
shared object is loaded on first use and a missing/unbuilt library raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  — must be imported BEFORE the .so is loaded: libavsd_hip.so then binds to the HIP
#                runtime torch already mapped (one runtime, one device context) instead of a second copy

from . import precision as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVSD_LIB_PATH") or os.path.join(_HERE, "libavsd_hip.so")   # override: A/B-testing a kernel build
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.environ.get("AVSD_LIB_PATH_F16") or os.path.join(_HERE, "libavsd_hip_f16.so")}

c_void_p, c_int, c_float, c_int64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class GemmDesc(C.Structure):
    """Mirror of `avsd_gemm_desc` (include/avsd.h)."""

    _fields_ = [
        ("A", c_void_p), ("A2", c_void_p), ("W", c_void_p), ("out", c_void_p),
        ("bias", c_void_p), ("rowvec", c_void_p), ("res1", c_void_p), ("res2", c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("lda2", C.c_int32), ("k_split", C.c_int32), ("ldw", C.c_int32),
        ("ldc", C.c_int32), ("ldr1", C.c_int32), ("ldr2", C.c_int32),
        ("rows_per_vec", C.c_int32), ("ldv", C.c_int32),
        ("alpha", C.c_float),
        ("mode", C.c_int32), ("flags", C.c_int32), ("batch", C.c_int32),
        ("batch_stride_a", C.c_int64), ("batch_stride_w", C.c_int64), ("batch_stride_out", C.c_int64),
        ("hw", C.c_int32), ("frames", C.c_int32), ("cseg", C.c_int32),
        ("hs", C.c_int32), ("ws", C.c_int32), ("ho", C.c_int32), ("wo", C.c_int32),
        ("cin", C.c_int32), ("stride", C.c_int32), ("ups", C.c_int32), ("pad", C.c_int32),
        ("tile", C.c_int32),
        ("split_k", C.c_int32), ("splitk_ws", c_void_p),
        ("rowstats", c_void_p), ("ln_stats", c_void_p), ("ln_colsum", c_void_p), ("ln_nblk", C.c_int32), ("ln_eps", C.c_float),
        ("out_master", c_void_p), ("ldm", C.c_int32), ("raster_g", C.c_int32),
        ("a_lo", C.c_int64), ("a2_lo", C.c_int64), ("w_lo", C.c_int64), ("out_lo", C.c_int64), ("res1_lo", C.c_int64),
        ("res2_lo", C.c_int64),
        ("stats_pos", c_void_p), ("ln_rowvec", c_void_p), ("pos_hw", C.c_int32), ("pos_frames", C.c_int32),
    ]


class XAttnDesc(C.Structure):
    """Mirror of `avsd_xattn_desc` (include/avsd.h)."""

    _fields_ = [
        ("h", c_void_p), ("ldh", C.c_int32), ("res_f32", C.c_int32),
        ("res", c_void_p), ("ldres", C.c_int32),
        ("M", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32), ("L", C.c_int32),
        ("ln_stats", c_void_p), ("ln_eps", C.c_float), ("scale", C.c_float),
        ("wq", c_void_p), ("ldwq", C.c_int32), ("lk", C.c_int32),
        ("q_colsum", c_void_p), ("q_bias", c_void_p),
        ("k", c_void_p), ("vt", c_void_p),
        ("lk_pad", C.c_int32), ("q_per_kv", C.c_int32),
        ("wo", c_void_p), ("ldwo", C.c_int32), ("ldo", C.c_int32),
        ("o_bias", c_void_p), ("out", c_void_p),
        ("out_master", c_void_p), ("ldm", C.c_int32), ("reserved0", C.c_int32),
        ("rowstats", c_void_p),
        ("stats_pos", c_void_p), ("pos_hw", C.c_int32), ("pos_frames", C.c_int32),
    ]


# name -> (restype, argtypes); exactly the symbols include/avsd.h declares
SIGNATURES = {
    "avsd_abi_version": (c_int, []),
    "avsd_precision": (C.c_char_p, []),
    "avsd_last_error": (C.c_char_p, []),
    "avsd_device_info": (c_int, [C.c_char_p, c_int, C.POINTER(c_int)]),
    "avsd_gemm_bf16": (c_int, [C.POINTER(GemmDesc), c_void_p]),
    "avsd_sizeof_gemm_desc": (c_int, []),
    "avsd_cross_attention_block": (c_int, [C.POINTER(XAttnDesc), c_void_p]),
    "avsd_gemm_conv3r_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "avsd_gemm_conv3r2d_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "avsd_cross_attention_block_supported": (c_int, [c_int, c_int, c_int]),
    "avsd_sizeof_xattn_desc": (c_int, []),
    "avsd_linear_small_m": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_groupnorm_stats": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "avsd_groupnorm_apply": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                     c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "avsd_ln_fold": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "avsd_groupnorm_fused_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "avsd_groupnorm_fused": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                     c_int, c_void_p, c_int, c_void_p]),
    "avsd_groupnorm_nchunks": (c_int, [c_int, c_int, c_int]),
    "avsd_groupnorm_scratch_floats": (c_int, [c_int, c_int, c_int, c_int]),
    "avsd_layernorm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    "avsd_softmax_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "avsd_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p]),
    "avsd_attention_fp8": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p]),
    "avsd_temporal_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "avsd_ncfhw_to_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "avsd_rows_to_ncfhw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "avsd_guided_step": (c_int, [c_void_p, c_int, c_float, c_float, c_void_p, c_int, c_float, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                 c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_vae_postprocess": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "avsd_vae_postprocess_u8": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "avsd_kaldi_fbank": (c_int, [c_void_p, c_int, c_int, C.c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                 c_void_p, c_int, c_float, c_float, c_void_p]),
    "avsd_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_vit_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_copy": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "avsd_xattn_pack_kv": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    # split-precision ("x2") storage: main + rest planes, three-pass MFMA products
    "avsd_linear_small_m_x2": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "avsd_groupnorm_stats_x2": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                        c_int, c_void_p]),
    "avsd_groupnorm_apply_x2": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "avsd_groupnorm_fused_x2": (c_int, [c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_float, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "avsd_layernorm_x2": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                  c_int, c_int, c_void_p]),
    "avsd_attention_x2": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p]),
    "avsd_temporal_attention_x2": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_int, c_float,
                                           c_void_p]),
    "avsd_softmax_rows_x2": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "avsd_ncfhw_to_rows_x2": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "avsd_split_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "avsd_vae_postprocess_x2": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "avsd_vae_postprocess_u8_x2": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p]),
    "avsd_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    # launch plans (asva_amd/plan.py records them; any host replays them)
    "avsd_plan_bundle_load": (c_int, [C.c_char_p, C.POINTER(c_void_p)]),
    "avsd_plan_bundle_free": (None, [c_void_p]),
    "avsd_plan_bundle_num_buffers": (c_int, [c_void_p]),
    "avsd_plan_bundle_buffer_bytes": (c_int64, [c_void_p, c_int]),
    "avsd_plan_bundle_bind": (c_int, [c_void_p, c_int, c_void_p]),
    "avsd_plan_bundle_num_regions": (c_int, [c_void_p]),
    "avsd_plan_bundle_region": (c_int, [c_void_p, c_int, C.POINTER(C.c_char_p), C.POINTER(c_int), C.POINTER(c_int64), C.POINTER(c_int64),
                                        C.POINTER(c_int)]),
    "avsd_plan_bundle_find_region": (c_int, [c_void_p, C.c_char_p]),
    "avsd_plan_bundle_num_plans": (c_int, [c_void_p]),
    "avsd_plan_bundle_plan_name": (C.c_char_p, [c_void_p, c_int]),
    "avsd_plan_bundle_find_plan": (c_int, [c_void_p, C.c_char_p]),
    "avsd_plan_num_calls": (c_int, [c_void_p, c_int]),
    "avsd_plan_run": (c_int, [c_void_p, c_int, c_void_p]),
    "avsd_unet_set_conditioning": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "avsd_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "avsd_vae_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "avsd_plan_region_ptr": (c_void_p, [c_void_p, C.c_char_p, C.POINTER(c_int64)]),
}

_libs = {}


class AvsdError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The kernel library of the current storage precision (asva_amd.precision), loaded once per precision.  Raises if
    it has not been built — there is no CPU path."""
    handle = _libs.get(P.NAME)
    if handle is None:
        path = LIB_PATHS[P.NAME]
        if not os.path.exists(path):
            raise AvsdError(
                f"{path} not found: build it with `python -m asva_amd.build` "
                "(hipcc --offload-arch=gfx950).  asva_amd has no fallback compute path."
            )
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.avsd_sizeof_gemm_desc() != C.sizeof(GemmDesc):
            raise AvsdError(
                f"avsd_gemm_desc layout mismatch: C {handle.avsd_sizeof_gemm_desc()} B vs ctypes {C.sizeof(GemmDesc)} B"
            )
        if handle.avsd_sizeof_xattn_desc() != C.sizeof(XAttnDesc):
            raise AvsdError(
                f"avsd_xattn_desc layout mismatch: C {handle.avsd_sizeof_xattn_desc()} B vs ctypes {C.sizeof(XAttnDesc)} B"
            )
        if handle.avsd_precision().decode() != P.NAME:
            raise AvsdError(f"{path} computes in {handle.avsd_precision().decode()}, expected {P.NAME}")
        _libs[P.NAME] = handle
    return handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().avsd_last_error()
        raise AvsdError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")

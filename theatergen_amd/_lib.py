"""ctypes binding of libtheatergen_hip.so (C ABI: include/theatergen_hip.h).

The library is the ONLY compute path: ``lib()`` raises RuntimeError if it is missing (no CPU / PyTorch
fallback).  Any non-zero return code becomes ``RuntimeError(tg_last_error())`` so callers keep the policy of
reference ``generate.py:250-259`` (RuntimeError => skip the turn).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# THEATERGEN_HIP_LIB: point at another build of the same C ABI (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("THEATERGEN_HIP_LIB") or os.path.join(HERE, "lib", "libtheatergen_hip.so")

TG_BF16, TG_F16 = 0, 1
ABI_VERSION = 307          # TG_ABI_VERSION of include/theatergen_hip.h this binding was written against
ACT_NONE, ACT_SILU, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2, 3

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("mode", i32), ("a0", vp), ("a1", vp), ("c0", i32), ("c1", i32),
        ("batch", i32), ("in_h", i32), ("in_w", i32), ("out_h", i32), ("out_w", i32),
        ("stride", i32), ("upsample", i32), ("w", vp), ("M", i64), ("N", i64), ("K", i64),
        ("bias", vp), ("bvec", vp), ("ldbvec", i64), ("rows_per_batch", i64), ("res", vp), ("ldres", i64),
        ("act", i32), ("geglu", i32), ("out_scale", f32), ("out", vp), ("ldc", i64), ("n_split", i64),
        ("out_t", vp), ("ldt", i64), ("workspace", vp), ("workspace_bytes", i64),
        ("force_split_k", i32), ("force_tile", i32), ("a_rows_per_batch", i64), ("a_batch_stride", i64),
        ("pad_mode", i32), ("a_silu", i32), ("a_coef", vp), ("ln_u", vp), ("ln_v", vp), ("ln_eps", f32), ("ln_rows", vp),
        ("lda", i64), ("ldw", i64), ("out_gn_partials", vp), ("out_gn_groups", i32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("batch", i32), ("heads", i32), ("head_dim", i32), ("n_q", i32),
        ("q", vp), ("q_ld", i64), ("q_bs", i64),
        ("k0", vp), ("k0_ld", i64), ("k0_bs", i64), ("vt0", vp), ("vt0_ld", i64), ("vt0_bs", i64), ("len0", i32),
        ("k1", vp), ("k1_ld", i64), ("k1_bs", i64), ("vt1", vp), ("vt1_ld", i64), ("vt1_bs", i64), ("len1", i32),
        ("scale", f32), ("w1", f32), ("out", vp), ("out_ld", i64), ("out_bs", i64), ("causal", i32), ("w1_dev", vp),
        ("mask", vp), ("mask_bs", i64), ("mask_hs", i64), ("mask_qs", i64),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("batch", i32), ("heads", i32), ("head_dim", i32), ("n", i32),
        ("q", vp), ("k", vp), ("v", vp), ("dout", vp), ("ld", i64), ("bs", i64),
        ("qt", vp), ("kt", vp), ("doutt", vp), ("t_ld", i64), ("t_bs", i64),
        ("stats", vp), ("dq", vp), ("dk", vp), ("dv", vp), ("scale", f32),
    ]


class AttnBwdCrossDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("batch", i32), ("heads", i32), ("head_dim", i32), ("n_q", i32), ("n_k", i32),
        ("q", vp), ("dout", vp), ("q_ld", i64), ("q_bs", i64),
        ("k", vp), ("v", vp), ("k_ld", i64), ("k_bs", i64),
        ("kt", vp), ("t_ld", i64), ("t_bs", i64),
        ("extra", vp), ("extra_ld", i64), ("stats", vp), ("dq", vp), ("scale", f32), ("ds_scale", f32),
    ]


class RcLinearDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("x", vp), ("ldx", i64), ("wpk", vp), ("res", vp), ("ldres", i64),
        ("out", vp), ("ldc", i64), ("M", i64), ("N", i32), ("K", i32), ("ln", i32), ("ln_eps", f32), ("variant", i32), ("v640", vp), ("u640", vp),
    ]


class RcXattnDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("h", vp), ("ldh", i64), ("wq", vp), ("kv", vp), ("wo", vp), ("out", vp), ("ldc", i64), ("M", i64),
        ("rows_per_batch", i32), ("text_len", i32), ("ip_tokens", i32), ("ln_eps", f32), ("ip_scale", vp),
    ]


class XqAttnDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("x", vp), ("ldx", i64), ("wq", vp), ("ln_u", vp), ("ln_v", vp), ("ln_eps", f32), ("kv", vp), ("ip_scale", vp),
        ("out", vp), ("ldc", i64), ("M", i64), ("C", i32), ("head_dim", i32), ("rows_per_batch", i32), ("text_len", i32), ("ip_tokens", i32),
    ]


class RcFfDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("h", vp), ("ldh", i64), ("w1", vp), ("w2", vp), ("b2", vp), ("wpo", vp), ("res0", vp), ("ldres", i64),
        ("out", vp), ("ldc", i64), ("M", i64), ("inner", i32), ("ln_eps", f32), ("dbg", i32),
    ]


class SkinnySeg(C.Structure):
    _fields_ = [("ptr", vp), ("ld", i64), ("batch_stride", i64), ("n_end", i32), ("transposed", i32)]


class SkinnyDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("x", vp), ("ldx", i64), ("wpk", vp), ("M", i64), ("N", i32), ("K", i32), ("ln", i32), ("ln_eps", f32), ("ln_u", vp), ("ln_v", vp),
        ("bias", vp), ("act", i32), ("res", vp), ("ldres", i64), ("nseg", i32), ("rows_per_batch", i32), ("seg", SkinnySeg * 3),
    ]


class RcFrontDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("x", vp), ("ldx", i64), ("coef", vp), ("win", vp), ("wqkv", vp), ("y", vp), ("ldy", i64), ("qk", vp), ("ldqk", i64),
        ("vt", vp), ("ldt", i64), ("M", i64), ("rows_per_batch", i32), ("ln_eps", f32), ("dbg", i32),
    ]


class GuidanceItem(C.Structure):
    _fields_ = [
        ("attn", vp), ("grad", vp), ("mask", vp), ("ref", vp),
        ("heads", i32), ("hw", i32), ("n_tok", i32), ("token", i32),
        ("kind", i32), ("k_fg", i32), ("k_bg", i32), ("reserved", i32),
        ("fg_w", f32), ("bg_w", f32), ("scale", f32), ("eps", f32),
    ]


class GuidancePItem(C.Structure):
    _fields_ = [
        ("attn_slot", i32), ("grad_slot", i32), ("mask_slot", i32), ("ref_slot", i32),
        ("heads", i32), ("hw", i32), ("n_tok", i32), ("token", i32),
        ("kind", i32), ("k_fg", i32), ("k_bg", i32), ("reserved", i32),
        ("fg_w", f32), ("bg_w", f32), ("scale", f32), ("eps", f32),
    ]


GUIDANCE_MAX_SLOTS = 64

# name -> (restype, argtypes); every symbol declared in include/theatergen_hip.h
SIGNATURES = {
    "tg_version": (i32, []),
    "tg_last_error": (C.c_char_p, []),
    "tg_gemm": (i32, [C.POINTER(GemmDesc), vp]),
    "tg_gemm_workspace_bytes": (i64, [C.POINTER(GemmDesc)]),
    "tg_gemm_plan": (i32, [C.POINTER(GemmDesc), vp, vp, vp, vp]),
    "tg_attention": (i32, [C.POINTER(AttnDesc), vp]),
    "tg_attention_bwd": (i32, [C.POINTER(AttnBwdDesc), vp]),
    "tg_attention_bwd_cross": (i32, [C.POINTER(AttnBwdCrossDesc), vp]),
    "tg_attn_probs": (i32, [i32, i32, i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, i32, f32, vp, i32, vp, vp]),
    "tg_groupnorm_scratch_bytes": (i64, [i32, i64, i32]),
    "tg_groupnorm": (i32, [i32, vp, vp, i32, i32, i32, i64, i32, f32, vp, vp, i32, vp, vp, vp]),
    "tg_layernorm": (i32, [i32, vp, i64, i32, i64, f32, vp, vp, vp, i64, vp]),
    "tg_layernorm_stats": (i32, [i32, vp, i64, i32, i64, f32, vp, vp]),
    "tg_groupnorm_coef": (i32, [i32, vp, vp, i32, i32, i32, i64, i32, f32, vp, vp, vp, vp, vp]),
    "tg_groupnorm_from_partials": (i32, [i32, vp, i32, i32, i64, i32, f32, vp, vp, i32, vp, vp, vp, i32, vp]),
    "tg_gemm_gn_partial_blocks": (i32, [C.POINTER(GemmDesc)]),
    "tg_geglu": (i32, [i32, vp, i64, i64, vp, vp]),
    "tg_act": (i32, [i32, vp, i64, i32, vp, vp]),
    "tg_add": (i32, [i32, vp, vp, i64, vp, vp]),
    "tg_transpose": (i32, [i32, vp, i32, i32, i32, vp, vp]),
    "tg_conv1x1_nchw": (i32, [vp, i32, i32, i32, i64, vp, vp, f32, vp, vp]),
    "tg_softmax_rows": (i32, [i32, vp, i64, i32, i64, f32, vp, i64, vp]),
    "tg_conv_in": (i32, [i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
    "tg_conv_out": (i32, [i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp]),
    "tg_conv_out_gn": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp]),
    "tg_conv_out_takes_coef": (i32, [i32, i32, i32, i32]),
    "tg_timestep_embedding": (i32, [i32, vp, vp, i32, i32, i32, i32, f32, vp, i64, vp]),
    "tg_step_epilogue": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, i32, vp]),
    "tg_blend_latents": (i32, [vp, vp, vp, i32, i32, f32, f32, i32, vp, vp]),
    "tg_shift": (i32, [vp, i64, i32, i32, i32, i32, vp, vp]),
    "tg_masked_compose": (i32, [vp, vp, vp, i64, i32, vp]),
    "tg_gaussian_sample": (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
    "tg_add_noise": (i32, [vp, vp, vp, vp, i32, i64, vp, vp]),
    "tg_guidance_topk": (i32, [vp, i32, i32, i32, i32, vp, i32, i32, f32, f32, f32, vp, vp, vp]),
    "tg_guidance_ratio": (i32, [vp, i32, i32, i32, i32, vp, f32, vp, vp, vp]),
    "tg_guidance_ref": (i32, [vp, i32, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp]),
    "tg_guidance_batch": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "tg_guidance_plan_run": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, vp]),
    "tg_groupnorm_bwd_scratch_bytes": (i64, [i32, i64, i32]),
    "tg_groupnorm_bwd": (i32, [i32, vp, vp, i32, i64, i32, i32, f32, vp, vp, i32, vp, vp, vp]),
    "tg_layernorm_bwd": (i32, [i32, vp, vp, i64, i32, f32, vp, vp, vp]),
    "tg_geglu_bwd": (i32, [i32, vp, vp, i64, i64, vp, vp]),
    "tg_softmax_bwd_rows": (i32, [i32, vp, i32, i64, vp, i64, vp, i64, i64, i32, f32, vp, vp, i64, vp]),
    "tg_sumpool2x2": (i32, [i32, vp, i32, i32, i32, i32, vp, vp]),
    "tg_rc_linear": (i32, [C.POINTER(RcLinearDesc), vp]),
    "tg_rc_xattn": (i32, [C.POINTER(RcXattnDesc), vp]),
    "tg_rc_kv_pack": (i32, [i32, i32, vp, vp, i64, i32, vp, vp, i64, i32, vp, vp]),
    "tg_rc_ff": (i32, [C.POINTER(RcFfDesc), vp]),
    "tg_rc_front": (i32, [C.POINTER(RcFrontDesc), vp]),
    "tg_skinny_gemm": (i32, [C.POINTER(SkinnyDesc), vp]),
    "tg_xq_attn": (i32, [C.POINTER(XqAttnDesc), vp]),
    "tg_xq_kv_bytes": (i64, [i32, i32, i32]),
    "tg_xq_kv_pack": (i32, [i32, i32, i32, i32, vp, vp, i64, i32, vp, vp, i64, i32, vp, vp]),
    "tg_debug_mfma32": (i32, [i32, vp, vp, vp, vp]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"theatergen_amd: HIP library not found at {LIB_PATH}; build it with "
                "`python -m theatergen_amd.build` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7) and this library needs the same
        # SONAME: torch must be loaded FIRST so both share ONE HIP runtime (streams, device memory).  Loaded the
        # other way round the process ends up with two runtimes and the second reports "no ROCm-capable device".
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        h.tg_version.restype = i32
        got = h.tg_version()
        # dev A/B of an OLDER build of the same ABI family (scripts/ab.py: old .so vs new .so on one box): descriptor fields are only ever
        # appended, so a library one revision behind reads a prefix of what this binding writes (entry points it lacks stay unbound: the arm
        # must not reach them, e.g. TG_GN_EPI=0).  Never the default path.
        compat = os.environ.get("THEATERGEN_HIP_LIB") and os.environ.get("THEATERGEN_HIP_ABI_COMPAT") == str(got)
        for name, (res, args) in SIGNATURES.items():
            if compat and not hasattr(h, name):
                continue
            fn = getattr(h, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if got != ABI_VERSION and not compat:
            raise RuntimeError(f"theatergen_amd: {LIB_PATH} has ABI version {got}, this binding needs {ABI_VERSION}: "
                               "rebuild with `python -m theatergen_amd.build` (a stale library would misread descriptors)")
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().tg_last_error()
        raise RuntimeError(f"theatergen_hip error {rc}: {msg.decode() if msg else ''}")

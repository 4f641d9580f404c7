"""ctypes binding of libvdn_hip.so (include/vdn.h). No CPU fallback: if the HIP library is missing
or cannot be loaded, importing this module raises with the build command."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("VDN_LIB") or os.path.join(PKG, "lib", "libvdn_hip.so")

F16, BF16, F32, NONE = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU = 0, 1, 2, 3
A_PLAIN, A_CONV3X3 = 0, 1
ST_PLAIN, ST_HEADS, ST_CONVT, ST_GEGLU = 0, 1, 2, 3
PACK_LINEAR, PACK_CONV3X3, PACK_CONV3X3_TAPS, PACK_CONVT, PACK_GEGLU, PACK_ROPE = 0, 1, 2, 3, 4, 5

i32, vp, fp = C.c_int32, C.c_void_p, C.c_void_p


class GemmDesc(C.Structure):
    """Mirror of vdn_gemm_desc (include/vdn.h) — field order and types must match exactly;
    the layout is checked against the library's own sizeof/offsetof probes at import (bottom of this file) and in tests/test_host.py."""
    _fields_ = [
        ("dt", i32), ("M", i32), ("N", i32), ("K", i32),
        ("A", vp), ("a_mode", i32), ("lda", i32), ("relu_a", i32),
        ("cB", i32), ("cH", i32), ("cW", i32), ("cC", i32), ("cOH", i32), ("cOW", i32), ("cstride", i32),
        ("W", vp), ("ldb", i32),
        ("bias", fp), ("rowadd", fp), ("act", i32), ("gamma", fp), ("tab", fp), ("tab_mod", i32), ("tab_off", i32),
        ("res1", vp), ("res1_dt", i32), ("ldr1", i32),
        ("res2", vp), ("res2_dt", i32), ("ldr2", i32),
        ("store", i32), ("out", vp), ("out_dt", i32), ("ldc", i32), ("row_group", i32), ("row_skip", i32),
        ("dst", vp * 3), ("nsplit", i32), ("heads", i32), ("tokens", i32), ("tok_off", i32), ("tpad", i32),
        ("transposed", i32 * 3), ("rope", i32 * 3),
        ("rope_cs", fp), ("rope_mod", i32),
        ("ck", i32), ("cout", i32), ("zeros", vp),
        ("A_lo", vp), ("W_lo", vp), ("out_lo", vp), ("dst_lo", vp * 3), ("res1_lo", vp), ("res2_lo", vp),
        ("conv_korder", i32), ("cu_hint", i32),
        ("splitk_ws", vp), ("splitk_ws_bytes", C.c_int64), ("ksplit", i32),
        ("dst8", vp * 3),
        ("A8", vp), ("W8", vp), ("out8", vp),
        ("a_kt", i32), ("w_kt", i32), ("out_kt", i32), ("x8_terms", i32),
        ("tuning", vp),
    ]


class GemmTuning(C.Structure):
    """Mirror of vdn_gemm_tuning (include/vdn.h): kernel-selection knobs of vdn_gemm, attached per launch (desc.tuning)."""
    _fields_ = [("force_bm", i32), ("p8", i32), ("no_splitk", i32), ("no_pipe", i32), ("splitk_p8", i32),
                ("cus", i32), ("splitk_occ", i32), ("splitk_max", i32), ("min_tiles", i32), ("f128", C.c_float),
                ("f192", C.c_float), ("x8", i32)]


class VdnError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (there is no CPU fallback). "
            f"Build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950).")
    try:
        return C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"cannot load {LIB_PATH}: {e}. Rebuild with `python __graft_entry__.py build`.") from e


lib = _load()

EXPORTS = {
    "vdn_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "vdn_gemm_get_tuning": (C.c_int, [C.POINTER(GemmTuning)]),
    "vdn_layernorm": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, fp, C.c_float, fp, C.c_int, C.c_int,
                                C.c_int, vp, vp, C.c_int, fp, vp, C.c_int, vp]),
    "vdn_flash_attn": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    "vdn_temporal_attn": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fp, vp]),
    "vdn_groupnorm": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, fp,
                                C.c_int, vp]),
    "vdn_upsample_bilinear": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_upsample_bilinear_f32": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_patchify": (C.c_int, [C.c_int, fp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_fill_row": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_bicubic": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]),
    "vdn_preprocess": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
    "vdn_add_vec": (C.c_int, [fp, fp, C.c_float, fp, C.c_int, C.c_int, vp]),
    "vdn_head_out": (C.c_int, [C.c_int, vp, vp, fp, C.c_float, fp, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_depth_tail": (C.c_int, [C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, fp, fp, C.c_float, fp,
                                 C.c_int, C.c_int, C.c_int, vp]),
    "vdn_pack_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vdn_pack_ldb": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vdn_pack_weight": (C.c_int, [C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "vdn_pack_bias": (C.c_int, [C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, vp]),
    "vdn_pack_x8": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "vdn_pack_x8_f32": (C.c_int, [fp, C.c_int, C.c_int, vp, vp, vp]),
    "vdn_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "vdn_groupnorm_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vdn_mask_down1": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, vp]),
    "vdn_mask_down2": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, vp]),
    "vdn_dwconv7": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, vp]),
    "vdn_addtab_cast": (C.c_int, [C.c_int, fp, fp, C.c_int, C.c_int, vp, vp, C.c_size_t, C.c_int, vp]),
    "vdn_cast": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_size_t, vp]),
    "vdn_temporal_attn_last": (C.c_int, [C.c_int, fp, C.c_size_t, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, fp, fp, fp,
                                         C.c_float, vp, vp, vp]),
    "vdn_frame_median_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vdn_frame_median": (C.c_int, [fp, C.c_int, C.c_size_t, fp, vp, vp]),
    "vdn_refine_scale": (C.c_int, [fp, fp, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, vp]),
    "vdn_refine_pack": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_refine_finish": (C.c_int, [fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, fp, C.c_size_t, vp]),
    "vdn_stitch_workspace_bytes": (C.c_size_t, []),
    "vdn_stitch_fit": (C.c_int, [fp, fp, C.c_size_t, vp, fp, vp]),
    "vdn_stitch_apply": (C.c_int, [fp, fp, fp, fp, fp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vdn_sizeof_gemm_desc": (C.c_size_t, []),
    "vdn_offsetof_gemm_zeros": (C.c_size_t, []),
    "vdn_offsetof_gemm_res2_lo": (C.c_size_t, []),
    "vdn_version": (C.c_char_p, []),
    "vdn_arch_ok": (C.c_int, []),
}

for _name, (_res, _args) in EXPORTS.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library is stale: rebuild
    _fn.restype = _res
    _fn.argtypes = _args

if (lib.vdn_sizeof_gemm_desc() != C.sizeof(GemmDesc) or lib.vdn_offsetof_gemm_zeros() != GemmDesc.zeros.offset
        or lib.vdn_offsetof_gemm_res2_lo() != GemmDesc.res2_lo.offset):
    raise ImportError("vdn_gemm_desc layout mismatch between include/vdn.h and vdn/_abi.py — rebuild the library")


# The library keeps no selection state: a launch carries its own knobs in desc.tuning. OVERRIDE is the Python host's
# (process-wide, test / tool oriented) choice of what Runtime.gemm attaches to every descriptor; None = the library defaults.
OVERRIDE = None


def default_tuning() -> GemmTuning:
    t = GemmTuning()
    check(lib.vdn_gemm_get_tuning(C.byref(t)), "vdn_gemm_get_tuning")
    return t


def get_tuning() -> GemmTuning:
    """The knobs Runtime.gemm launches with: the override if one is set, else the library's environment-derived defaults."""
    t = GemmTuning()
    C.memmove(C.byref(t), C.byref(OVERRIDE if OVERRIDE is not None else default_tuning()), C.sizeof(GemmTuning))
    return t


def set_tuning(**kw):
    """Change some vdn_gemm_tuning fields for every later Runtime.gemm launch of this process; returns the previous state
    (restore with restore_tuning)."""
    global OVERRIDE
    old, new = OVERRIDE, get_tuning()
    for k, v in kw.items():
        if not hasattr(new, k):
            raise AttributeError(k)
        setattr(new, k, v)
    OVERRIDE = new
    return old


def restore_tuning(t):
    global OVERRIDE
    OVERRIDE = t


def check(rc: int, what: str):
    if rc != 0:
        if rc <= -1000:
            raise VdnError(f"{what}: HIP error {-rc - 1000}")
        raise VdnError(f"{what}: rejected arguments (vdn_status {rc})")

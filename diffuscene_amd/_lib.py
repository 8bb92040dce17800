"""ctypes binding of libdiffuscene_hip.so (C ABI declared in include/diffuscene_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).  There is no
fallback: if the shared object is missing, ``load()`` raises, and every op in this package goes through it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffuscene_hip.so")
# DSC_HIP_LIB selects another build of the SAME library (e.g. the host-ASAN build of `python __graft_entry__.py --asan`);
# it is still the HIP extension -- there is no fallback of any kind behind this switch.
if os.environ.get("DSC_HIP_LIB"):
    LIB_PATH = os.path.abspath(os.environ["DSC_HIP_LIB"])

ACT_NONE, ACT_GELU, ACT_SILU, ACT_LEAKY01 = 0, 1, 2, 3
SS_NONE, SS_PER_TOKEN, SS_PER_SCENE, SS_PER_SLOT, SS_BY_INDEX = 0, 1, 2, 3, 4
MEAN_EPS, MEAN_X0, MEAN_V = 0, 1, 2
TILE_GN_80_W8, TILE_GN_80_W4, TILE_160x256, TILE_160x128_W4 = 1, 2, 5, 8      # DSC_TILE_* (dsc_gemm_split_tile)
TILE_WAVE_GN, TILE_WAVE_DENSE, TILE_WAVE_GN_64 = 10, 11, 12
GEMM_ROW_INVARIANT = 1          # dsc_gemm_args.flags (include/diffuscene_hip.h)
WS_MAX = 64
MAX_TOKENS_PER_SCENE = 160

_ERR = {-1: "DSC_EINVAL (bad shape / null pointer)", -2: "DSC_EALIGN (16-byte alignment required)",
        -3: "DSC_ERANGE (size outside kernel limits)"}

c_f32p = C.c_void_p
c_i64p = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("a1", C.c_void_p), ("lda1", C.c_int64), ("k1", C.c_int32),
        ("a2", C.c_void_p), ("lda2", C.c_int64), ("k2", C.c_int32),
        ("w", C.c_void_p), ("ldw", C.c_int64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("y", C.c_void_p), ("ldy", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32),
        ("act_in", C.c_int32), ("act_out", C.c_int32),
        ("batch", C.c_int32),
        ("sa1", C.c_int64), ("sa2", C.c_int64), ("sw", C.c_int64), ("sbias", C.c_int64),
        ("sres", C.c_int64), ("sy", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
        ("tokens_per_scene", C.c_int32),
        ("scale_shift", C.c_void_p), ("ld_ss", C.c_int64), ("ss_mode", C.c_int32),
        ("preact", C.c_void_p), ("ld_preact", C.c_int64),
        ("ss_index", C.c_void_p),
        ("w_planes", C.c_void_p),
        ("actgrad_x", C.c_void_p), ("ld_actgrad", C.c_int64),
        ("ss_rows", C.c_int32),
        ("w_planes_layout", C.c_int32),
        ("gnb_z", C.c_void_p), ("ld_gnb_z", C.c_int64),
        ("gnb_dgamma", C.c_void_p), ("gnb_dbeta", C.c_void_p), ("gnb_dbias", C.c_void_p), ("gnb_pstride", C.c_int64),
        ("gnb_dss", C.c_void_p), ("ld_gnb_dss", C.c_int64),
        ("flags", C.c_int32),
    ]


class SplitItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("ldw", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("planes", C.c_void_p),
                ("transpose", C.c_int32)]


class SmallKItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("k_in", C.c_int32), ("w", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("ldy", C.c_int64)]


class ColSpan(C.Structure):
    _fields_ = [("src_col", C.c_int32), ("dst_col", C.c_int32), ("width", C.c_int32)]


SMALLK_MAX = 4


class WsItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class TnGroup(C.Structure):
    _fields_ = [("a1", C.c_void_p), ("lda1", C.c_int64), ("k1", C.c_int32),
                ("a2", C.c_void_p), ("lda2", C.c_int64), ("k2", C.c_int32),
                ("dy", C.c_void_p), ("ldd", C.c_int64),
                ("out", C.c_void_p), ("ldo", C.c_int64),
                ("dbias", C.c_void_p),
                ("m", C.c_int32), ("n", C.c_int32), ("kvalid", C.c_int32),
                ("tile0", C.c_int32),
                ("ws_offset", C.c_int64),
                ("tile0s", C.c_int32)]


class ColsumItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("m", C.c_int32), ("n", C.c_int32), ("out", C.c_void_p)]


class WsBwdItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dw_std", C.c_void_p), ("dw", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


# name -> (restype, argtypes); must list every function declared in include/diffuscene_hip.h
SIGNATURES = {
    "dsc_version": (C.c_int, []),
    "dsc_gemm_f32": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "dsc_gemm_gn_silu_f32": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "dsc_split_bf16x3_f32": (C.c_int, [C.POINTER(SplitItem), C.c_int32, C.c_void_p]),
    "dsc_gemm_arithmetic": (C.c_int, [C.POINTER(GemmArgs), C.c_int32]),
    "dsc_gemm_split_tile": (C.c_int, [C.POINTER(GemmArgs), C.c_int32]),
    "dsc_gemm_planes_layout": (C.c_int, [C.POINTER(GemmArgs), C.c_int32]),
    "dsc_get_split_wave": (C.c_int, []),
    "dsc_set_split_wave": (C.c_int, [C.c_int32]),
    "dsc_gemm_skinny": (C.c_int, [C.POINTER(GemmArgs), C.c_int32]),
    "dsc_get_skinny": (C.c_int, []),
    "dsc_set_skinny": (C.c_int, [C.c_int32]),
    "dsc_get_gemm_arithmetic": (C.c_int, []),
    "dsc_set_gemm_arithmetic": (C.c_int, [C.c_int32]),
    "dsc_gemm_layernorm_f32": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "dsc_gemm_splitk_f32": (C.c_int, [C.POINTER(GemmArgs), C.c_int32, c_f32p, C.c_int64, C.c_void_p]),
    "dsc_linear_smallk_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_linear_smallk_grouped_f32": (C.c_int, [C.POINTER(SmallKItem), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_gather_columns_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.POINTER(ColSpan), C.c_int32, C.c_void_p]),
    "dsc_weight_standardize_f32": (C.c_int, [C.POINTER(WsItem), C.c_int32, C.c_float, C.c_void_p]),
    "dsc_layernorm_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32,
                                    C.c_int32, C.c_float, C.c_void_p]),
    "dsc_linear_attention_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "dsc_attention_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64,
                                    C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "dsc_time_embedding_f32": (C.c_int, [c_i64p, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, c_f32p, C.c_void_p]),
    "dsc_activation_f32": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_void_p]),
    "dsc_q_sample_f32": (C.c_int, [c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int64, C.c_int32,
                                   C.c_void_p]),
    "dsc_p_sample_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                   c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "dsc_device_error_count": (C.c_int64, [C.c_int32]),
    "dsc_add_scalar_i64": (C.c_int, [c_i64p, C.c_int32, C.c_int64, C.c_void_p]),
    "dsc_knn16_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "dsc_rowsq_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_int64, c_f32p, C.c_void_p]),
    "dsc_knn_cov_f32": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, c_f32p, C.c_int64, C.c_void_p]),
    "dsc_gather_max_f32": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int64,
                                     C.c_void_p, C.c_void_p]),
    "dsc_gather_max_bwd_f32": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                         C.c_int64, C.c_void_p]),
    "dsc_bn_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "dsc_batchnorm_fwd_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                        C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64,
                                        C.c_void_p]),
    "dsc_batchnorm_eval_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32,
                                         c_f32p, C.c_void_p]),
    "dsc_batchnorm_bwd_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_int32, c_f32p,
                                        c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "dsc_rowmax_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_void_p, C.c_void_p]),
    "dsc_rowmax_bwd_f32": (C.c_int, [c_f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_void_p]),
    "dsc_point_affine_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, c_f32p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, c_f32p, C.c_void_p]),
    "dsc_point_affine_bwd_f32": (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p,
                                           C.c_int64, C.c_void_p]),
    "dsc_postfilter_compact_f32": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                             C.c_void_p, C.c_void_p]),
    "dsc_gemm_tn_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, C.c_int32, c_f32p, C.c_int64, c_f32p,
                                  C.c_int64, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int64, C.c_void_p]),
    "dsc_gemm_tn_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "dsc_gemm_tn_grouped_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int64, C.c_int64,
                                          C.c_void_p]),
    "dsc_gemm_tn_grouped_split_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, c_f32p, C.c_int64,
                                                C.c_int64, C.c_int32, C.c_void_p]),
    "dsc_set_tn_split_form": (C.c_int, [C.c_int32]),
    "dsc_get_tn_split_form": (C.c_int, []),
    "dsc_colsum_f32": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "dsc_colsum_grouped_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_gn_silu_bwd_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32,
                                      c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "dsc_weight_standardize_bwd_f32": (C.c_int, [C.POINTER(WsBwdItem), C.c_int32, C.c_float, C.c_void_p]),
    "dsc_layernorm_bwd_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "dsc_linear_attention_bwd_f32": (C.c_int, [c_f32p, C.c_int64] * 7 + [C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                                                          C.c_void_p]),
    "dsc_attention_bwd_f32": (C.c_int, [c_f32p, C.c_int64] * 7 + [C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "dsc_retrieve_nearest_f32": (C.c_int, [c_f32p, C.c_void_p, C.c_void_p, c_f32p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, c_f32p, C.c_void_p]),
    "dsc_encode_scene_batch_f32": (C.c_int, [c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i64p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double), c_f32p,
                                             C.c_int64, c_i64p, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_chamfer3d_forward_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p]),
    "dsc_chamfer3d_backward_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, c_f32p, c_f32p,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_grad_sumsq_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dsc_clip_coef_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    "dsc_adam_step_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    c_f32p, C.c_void_p]),
    "dsc_ddpm_loss_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_float),
                                    c_f32p, c_f32p, c_f32p] + [C.c_int32] * 12 + [C.c_float, C.c_int32, C.c_void_p]),
    "dsc_copy2d_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_add2d_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_activation_bwd_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_void_p]),
    "dsc_transpose_batched_f32": (C.c_int, [C.POINTER(WsItem), C.c_int32, C.c_void_p]),
    "dsc_transpose_f32": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dsc_complete_overwrite_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "diffuscene_amd has no CPU or PyTorch fallback." % LIB_PATH)
    # PyTorch-ROCm wheels carry their own libamdhip64; it must be in the process before our library is dlopen'ed,
    # otherwise our DT_NEEDED resolves to /opt/rocm's copy and the process ends up with two HIP runtimes (kernels
    # registered in one, streams created by the other -> hipErrorNoDevice on the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing("symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.dsc_get_gemm_arithmetic() < 0:
        raise ValueError("DSC_GEMM=%r: must be 'split' (default) or 'f32'" % os.environ.get("DSC_GEMM"))
    if lib.dsc_get_tn_split_form() < 0:
        raise ValueError("DSC_TN_FORM=%r: must be '2' (default), '1' or '0'" % os.environ.get("DSC_TN_FORM"))
    if lib.dsc_get_split_wave() < 0:
        raise ValueError("DSC_WAVE=%r: must be '1' / 'auto' (default) or '0'" % os.environ.get("DSC_WAVE"))
    _lib = lib
    return lib


def split_enabled():
    """The library's GEMM arithmetic switch (dsc_get_gemm_arithmetic): True = split-bf16 wherever a launch qualifies, False = exact-f32
    MFMA everywhere.  The ONLY place host code learns the arithmetic from -- nothing in the package parses DSC_GEMM."""
    return load().dsc_get_gemm_arithmetic() == 1


def gemm_mode():
    """What plans, engines and graphs bake in about the GEMM kernels: 0 = exact-f32 MFMA everywhere, 1 = split-bf16 on the block-staged
    kernel only, 2 = split-bf16 with the wave-autonomous kernel wherever a launch qualifies (default).  They are keyed by it and
    rebuilt on their next use after a switch."""
    lib = load()
    if lib.dsc_get_gemm_arithmetic() != 1:
        return 0
    return (2 if lib.dsc_get_split_wave() == 1 else 1) + 10 * lib.dsc_get_tn_split_form()     # (+ the weight-gradient tile form: plans bake its block map in)


def set_split_wave(on):
    """Switch the wave-autonomous kernel family of the split arithmetic (dsc_set_split_wave); returns the previous setting.  Both
    families compute bit-identical results."""
    prev = load().dsc_get_split_wave() == 1
    check(load().dsc_set_split_wave(1 if on else 0), "dsc_set_split_wave")
    return prev


def set_gemm_arithmetic(name):
    """Switch the arithmetic per call ('split' | 'f32'); returns the previous name.  Engines, training plans and sampling graphs are
    keyed by the arithmetic they were built under and are rebuilt on their next use after a switch."""
    if name not in ("split", "f32"):
        raise ValueError("gemm arithmetic must be 'split' or 'f32', got %r" % (name,))
    prev = "split" if split_enabled() else "f32"
    check(load().dsc_set_gemm_arithmetic(1 if name == "split" else 0), "dsc_set_gemm_arithmetic")
    return prev


def device_error_count(reset=False):
    """Out-of-range device timesteps the DDPM kernels clamped since the last reset (dsc_device_error_count; synchronises).  0 in a
    correct run: tests assert it, debugging sessions read it."""
    n = load().dsc_device_error_count(1 if reset else 0)
    if n < 0:
        raise RuntimeError("dsc_device_error_count failed (HIP error)")
    return int(n)


def check_indices(where):
    """DSC_CHECK_INDICES=1 (debugging): raise if a DDPM kernel had to clamp an out-of-range device timestep since the last check -- the
    reference's gather would have raised an index error there; the kernels clamp (memory-safe) and count.  Synchronises the device."""
    if os.environ.get("DSC_CHECK_INDICES", "0") == "1":
        n = device_error_count(reset=True)
        if n:
            raise IndexError("%s: %d out-of-range device timestep(s) were clamped into the schedule tables" % (where, n))


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))


def fn(name):
    return getattr(load(), name)

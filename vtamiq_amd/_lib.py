"""ctypes binding of libvtamiq_hip.so (include/vtamiq_hip.h).  No CPU fallback: if the library is missing or
does not load, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VTQ_LIB_PATH") or os.path.join(_HERE, "libvtamiq_hip.so")   # override: kernel A/B builds in tools/
# the fp8 experiment's build of the same sources (python -m vtamiq_amd.build --fp8; include/vtamiq_hip_fp8.h).  It is a second library with its
# own handle: load_fp8() -- nothing has to be exported before the import, and the product library stays the one every other path uses.
LIB_PATH_FP8 = os.environ.get("VTQ_LIB_PATH_FP8") or os.path.join(_HERE, "libvtamiq_hip_fp8.so")

PREC_BF16 = 0
PREC_BF16X3 = 1
PREC_FP16 = 2
PREC_FP16X3 = 3
PREC_FP16X2 = 4
PREC_FP8 = 5
PRECISIONS = {"bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "fp16": PREC_FP16, "fp16x3": PREC_FP16X3, "fp16x2": PREC_FP16X2, "fp8": PREC_FP8}
# operand-format codes of the per-kernel entry points (VTQ_NUM_*): MFMAs per product + 16 for fp16 planes
NUM = {"bf16": 1, "bf16x3": 3, "fp16": 17, "fp16x2": 18, "fp16x3": 19, "fp8": 33}
# MFMAs per product of the linear layers / of attention, per precision (bench.py)
# (fp8: one e4m3 MFMA per product at twice the 16-bit rate = 0.5 bf16-MFMA-equivalents)
MFMA_TERMS = {"bf16": (1, 1), "bf16x3": (3, 3), "fp16": (1, 1), "fp16x3": (3, 3), "fp16x2": (2, 3), "fp8": (0.5, 1)}
ABI_VERSION = 9
OPT_FULL_LAST_LAYER = 1
OPT_FP8_STATIC_SCALES = 2
OPT_FUSED_LAYERNORM = 4

KERNEL_CLASSES = ["convert", "patch_embed", "layernorm", "qkv", "attention", "out_proj", "fc1", "fc2", "head"]


class VtqConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hidden_size", "mlp_dim", "num_heads", "num_layers", "patch_dim", "pos_grid", "num_extra_tokens", "num_scales",
        "use_layer_scale", "calibrate", "diff_scale", "num_rgs", "num_rcabs", "ca_hidden", "precision", "num_adapters", "options")] + [
        ("reserved", C.c_int32 * 3)]


class VtqTensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


# name -> (restype, argtypes): every symbol include/vtamiq_hip.h declares
SIGNATURES = {
    "vtq_abi_version": (C.c_int, []),
    "vtq_last_error": (C.c_char_p, []),
    "vtq_create": (C.c_int, [C.POINTER(VtqConfig), C.POINTER(C.c_void_p)]),
    "vtq_destroy": (None, [C.c_void_p]),
    "vtq_load_weights": (C.c_int, [C.c_void_p, C.POINTER(VtqTensorDesc), C.c_int32, C.c_void_p]),
    "vtq_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "vtq_reserve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "vtq_forward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_forward_tokens": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_forward_pairwise": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_forward_pairwise_tokens": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_set_token_trace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtq_set_iqa_token": (C.c_int, [C.c_void_p, C.c_int32]),
    "vtq_debug_stop_after": (C.c_int, [C.c_void_p, C.c_int32]),
    "vtq_debug_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vtq_debug_gemm_diag": (C.c_int, [C.c_void_p, C.c_int32]),
    "vtq_debug_attention_variant": (C.c_int, [C.c_int32]),
    "vtq_debug_attention_map": (C.c_int, [C.c_int32]),
    "vtq_debug_cu_partition": (C.c_int, [C.c_int32, C.c_int32]),
    "vtq_debug_cu_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_attention_rule": (C.c_int, [C.c_int32] * 5),
    "vtq_debug_gemm_variant": (C.c_int, [C.c_int32]),
    "vtq_k_gemm_tile_rule": (C.c_int, [C.c_int32] * 4),
    "vtq_debug_mfma_stream": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "vtq_profile_enable": (C.c_int, [C.c_void_p, C.c_uint32]),
    "vtq_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "vtq_input_errors": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "vtq_k_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_gemm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                             C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                             C.c_void_p]),
    "vtq_k_gemm_rowln": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vtq_k_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_skinny_linear": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_k_diffnet_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_k_image_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_void_p]),
    "vtq_k_avgpool2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_gather_patches": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_repeat_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_rank_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtq_k_gemm_schedule": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "vtq_k_attention": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p]),
}

# the fp8 experiment (include/vtamiq_hip_fp8.h): exported only by a library built with -DVTQ_WITH_FP8 (python -m vtamiq_amd.build --fp8 ->
# LIB_PATH_FP8, load_fp8()); bound when present
FP8_SIGNATURES = {
    "vtq_fp8_calibrate": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "vtq_fp8_get_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
    "vtq_fp8_set_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
    "vtq_k_quant_rows_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "vtq_k_quant_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "vtq_k_gemm_fp8": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "vtq_fp8_reset": (C.c_int, [C.c_void_p]),
}

_libs = {}            # absolute path -> bound CDLL: one handle per library file (RTLD_LOCAL: two builds of the same sources coexist)


def load(path: str | None = None) -> C.CDLL:
    """Load (once per path) and bind an engine library -- the product library by default; RuntimeError (never a silent fallback) when it
    is absent or broken."""
    path = os.path.abspath(path or LIB_PATH)
    lib = _libs.get(path)
    if lib is not None:
        return lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP engine is not built.  Run `python -m vtamiq_amd.build{' --fp8' if path == os.path.abspath(LIB_PATH_FP8) else ''}` "
            "(there is no CPU fallback on the product path).")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise RuntimeError(f"failed to load {path}: {e}") from e
    # VTQ_LIB_PATH selects WHICH build is loaded (tools/build_abl.sh variants of this tree); it does not relax any check.  An A/B
    # build of an OLDER tree (different ABI: argument layouts may differ) loads only with VTQ_ALLOW_ABI_MISMATCH=1, with a warning.
    relaxed = os.environ.get("VTQ_ALLOW_ABI_MISMATCH") == "1"
    for name, (res, args) in SIGNATURES.items():
        if relaxed and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "vtq_fp8_calibrate"):
        for name, (res, args) in FP8_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if lib.vtq_abi_version() != ABI_VERSION:
        if not relaxed:
            raise RuntimeError(f"{path}: ABI version {lib.vtq_abi_version()}, this package binds version {ABI_VERSION} "
                               "(rebuild: python -m vtamiq_amd.build)")
        import warnings
        warnings.warn(f"{path}: ABI version {lib.vtq_abi_version()} != {ABI_VERSION} accepted (VTQ_ALLOW_ABI_MISMATCH=1): "
                      "struct / argument layouts may differ -- measurement use only")
    _libs[path] = lib
    return lib


def load_fp8() -> C.CDLL:
    """The fp8 experiment's library (LIB_PATH_FP8) through its own handle; RuntimeError when it is not built or is not an fp8 build."""
    lib = load(LIB_PATH_FP8)
    if not has_fp8(lib):
        raise RuntimeError(f"{LIB_PATH_FP8} was built without the fp8 experiment: `python -m vtamiq_amd.build --fp8 --force`")
    return lib


def has_fp8(lib: C.CDLL | None = None) -> bool:
    """Is `lib` (default: the product library) a build of the fp8 experiment (-DVTQ_WITH_FP8)?  The product library is not."""
    return hasattr(lib if lib is not None else load(), "vtq_fp8_calibrate")


def fp8_available() -> bool:
    """Does the fp8 experiment's library exist and load (tests: run or skip)?"""
    try:
        load_fp8()
        return True
    except RuntimeError:
        return False


def check(rc: int, lib: C.CDLL | None = None) -> None:
    """Raise the library's error text for a non-zero status (`lib`: the handle the call went through; default the product library)."""
    if rc != 0:
        raise RuntimeError("vtamiq_hip: " + (lib if lib is not None else load()).vtq_last_error().decode(errors="replace"))

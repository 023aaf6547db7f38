"""Helpers for the -m gpu tests: call the C ABI with torch device tensors."""

import torch

from vtamiq_amd import _lib

# operand formats of the dense contractions (include/vtamiq_hip.h VTQ_NUM_*): name -> (fp16?, MFMAs per product)
FORMATS = {"bf16": (0, 1), "bf16x3": (0, 3), "fp16": (1, 1), "fp16x2": (1, 2), "fp16x3": (1, 3)}


def stream():
    return torch.cuda.current_stream().cuda_stream


def num_code(fmt: str) -> int:
    return _lib.NUM[fmt]


def planes_of(fmt: str, role: str) -> int:
    """planes of an activation ('a') or weight ('w') tensor in format `fmt`"""
    f16, terms = FORMATS[fmt]
    if terms == 1:
        return 1
    return 2 if (role == "a" or terms == 3) else 1


def elt_dtype(fmt: str):
    return torch.float16 if FORMATS[fmt][0] else torch.bfloat16


def to_planes(x: torch.Tensor, fmt: str, role: str = "a"):
    """fp32 [..] -> 16-bit tensor [planes, ..] via the engine's own split kernel."""
    lib = _lib.load()
    npl = planes_of(fmt, role)
    x = x.contiguous().float()
    out = torch.empty((npl,) + tuple(x.shape), dtype=elt_dtype(fmt), device=x.device)
    _lib.check(lib.vtq_k_split(x.data_ptr(), out.data_ptr(), x.numel(), x.numel(), FORMATS[fmt][0], npl, stream()))
    return out


def planes_value(p: torch.Tensor) -> torch.Tensor:
    """hi (+ lo) as float64."""
    v = p[0].double()
    if p.shape[0] == 2:
        v = v + p[1].double()
    return v

"""Helpers for the -m gpu tests: call the C ABI with torch device tensors."""
import ctypes as C

import torch

from vtamiq_amd import _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_planes(x: torch.Tensor, nsplit: int):
    """fp32 [..] -> bf16 tensor [npl, ..] via the engine's own split kernel."""
    lib = _lib.load()
    npl = 1 if nsplit == 1 else 2
    x = x.contiguous().float()
    out = torch.empty((npl,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.vtq_k_split_bf16(x.data_ptr(), out.data_ptr(), x.numel(), x.numel(), nsplit, stream()))
    return out


def planes_value(p: torch.Tensor) -> torch.Tensor:
    """hi (+ lo) as float64."""
    v = p[0].double()
    if p.shape[0] == 2:
        v = v + p[1].double()
    return v

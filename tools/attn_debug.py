#!/usr/bin/env python3
"""Localise the attention error of a format: inputs exactly representable in one plane (lo planes zero) vs full hi/lo inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream
lib = _lib.load()
nseq, S, H = 4, 501, 768
rows = nseq * S + 128
g = torch.Generator(device="cpu").manual_seed(16)
base = (torch.randn(rows, 3 * H, generator=g) * 1.5)
def run(qkv, fmt):
    P = to_planes(qkv.cuda(), fmt, "a")
    out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device="cuda")
    _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
    torch.cuda.synchronize()
    nh = H // 64
    x = planes_value(P)[: nseq * S].view(nseq, S, 3, nh, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(nseq, S, H)
    got = planes_value(out)[: nseq * S].view(nseq, S, H)
    d = (got - ref).abs()
    return (d.max() / ref.abs().max()).item(), (d.mean() / ref.abs().mean()).item(), ((got - ref).mean() / ref.abs().mean()).item()
for scale in (1.5, 0.5, 0.1):
    x = base * (scale / 1.5)
    for fmt in ("bf16x3", "fp16x3", "fp16"):
        dt = elt_dtype(fmt)
        print(f"scale {scale} {fmt:7s} full inputs: max/mean/bias rel err %.2e %.2e %+.2e" % run(x, fmt),
              "| one-plane inputs: %.2e %.2e %+.2e" % run(x.to(dt).float(), fmt), flush=True)

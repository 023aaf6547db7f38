#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernel at the encoder shape (GPU box)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--nseq", type=int, default=64)
ap.add_argument("--S", type=int, default=501)
ap.add_argument("--H", type=int, default=768)
ap.add_argument("--fmt", nargs="+", default=["bf16", "bf16x3", "fp16", "fp16x3"])
ap.add_argument("--variant", type=int, default=-1, help="-1 the library's rule, 0 4-wave kernel, 1 pipelined kernel, 2 split")
a = ap.parse_args()
lib = _lib.load()
lib.vtq_debug_attention_variant(a.variant)
S_pad = a.S          # the engine packs sequences back to back
rows = a.nseq * S_pad + 128
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(rows, 3 * a.H, generator=g) * 1.5).cuda()
for fmt in a.fmt:
    P = to_planes(qkv, fmt, "a")
    out = torch.zeros((P.shape[0], rows, a.H), dtype=elt_dtype(fmt), device="cuda")
    call = lambda: _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * a.H, out.data_ptr(), rows * a.H, a.nseq, a.S, S_pad, a.H, num_code(fmt), stream()))
    call(); torch.cuda.synchronize()
    nh = a.H // 64
    x = planes_value(P)[: 2 * S_pad].view(2, S_pad, 3, nh, 64)[:, :a.S]
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(2, a.S, a.H)
    got = planes_value(out)[: 2 * S_pad].view(2, S_pad, a.H)[:, :a.S]
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ts.sort()
    fl = 4.0 * a.nseq * nh * a.S * a.S * 64
    print(f"attention {fmt} variant={a.variant} nseq={a.nseq} S={a.S} H={a.H}: {ts[3]*1e3:.1f} us  {fl/ts[3]/1e9:.1f} TF algorithmic  err {err:.1e}")

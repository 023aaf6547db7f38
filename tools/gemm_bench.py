#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel on the encoder shapes (GPU box).  Interleaved rounds, median time, and a
correctness check against an fp64 reference on a row sample each run.  VTQ_GEMM_FLAGS (kernels.h) selects ablations."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32256)       # 64 sequences x 501 tokens padded to the tile height
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--fmt", nargs="+", default=["bf16", "bf16x3", "fp16x2"])
ap.add_argument("--only", nargs="+", default=None)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
M = a.M
shapes = [("qkv", 2304, 768, 0), ("outproj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
g = torch.Generator(device="cpu").manual_seed(0)
tag = os.environ.get("VTQ_GEMM_FLAGS", "0")
for fmt in a.fmt:
    terms = FORMATS[fmt][1]
    for name, N, K, epi in shapes:
        if a.only and name not in a.only:
            continue
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        gamma = torch.randn(N, generator=g).to(dev)
        Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
        npl = Ap.shape[0]
        out = torch.zeros((npl, M, N), dtype=elt_dtype(fmt), device=dev) if epi != 2 else None
        x0 = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
        x = x0.clone() if epi == 2 else None

        def call():
            _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), epi, bias.data_ptr(),
                                      gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                      out.data_ptr() if epi != 2 else None, M * N, N, stream()))
        call()
        torch.cuda.synchronize()
        # correctness on sampled rows
        rows = torch.tensor([0, 1, 17, 255, 256, 1000, M - 129, M - 1], device=dev)
        h = planes_value(Ap)[rows] @ planes_value(Wp).t() + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(h)
        elif epi == 2:
            ref = x0[rows].double() + gamma.double() * h
        else:
            ref = h
        got = x[rows].double() if epi == 2 else planes_value(out)[rows]
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        tol = ({"bf16": 1e-2, "fp16": 2e-3}.get(fmt, 1e-4)) if epi != 2 else 3e-5
        ok = "ok" if err < tol else "WRONG"
        if tag != "0" and int(tag) & 1:
            ok = "(rows wrapped: values not checked)"
        ts = []
        for r in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        ts.sort()
        med = ts[len(ts) // 2]
        tf = 2.0 * M * N * K / (med * 1e-3) / 1e12
        print(f"flags={tag} {fmt:7s} {name:8s} M={M} N={N} K={K}: {med*1e3:8.1f} us  {tf:7.1f} TF algorithmic ({tf/2516.6:.3f} of bf16 peak; "
              f"x{terms} MFMA issue {tf*terms:7.1f})  min {ts[0]*1e3:.1f} us  err {err:.1e} {ok}", flush=True)
        del A, W, Ap, Wp, out, x, x0

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
def bench_fp8(name, N, K, epi):
    """e4m3 operands on the MX-scaled MFMA (vtq_k_gemm_fp8): epilogue 0 -> fp16 hi/lo planes, 1 -> e4m3 GELU, 2 -> residual"""
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    bias, gamma = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    a8 = torch.empty(M, K, dtype=torch.uint8, device=dev)
    w8 = torch.empty(N, K, dtype=torch.uint8, device=dev)
    inv = torch.empty(N, dtype=torch.float32, device=dev)
    _lib.check(lib.vtq_k_quant_fp8(A.data_ptr(), a8.data_ptr(), A.numel(), 8.0, stream()))
    _lib.check(lib.vtq_k_quant_rows_fp8(W.data_ptr(), w8.data_ptr(), inv.data_ptr(), N, K, stream()))
    out = (torch.zeros(1, M, N, dtype=torch.float16, device=dev) if epi == 0 else torch.zeros(M, N, dtype=torch.uint8, device=dev)) if epi != 2 else None
    x0 = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    x = x0.clone() if epi == 2 else None

    def call():
        _lib.check(lib.vtq_k_gemm_fp8(a8.data_ptr(), K, w8.data_ptr(), inv.data_ptr(), 1.0 / 8.0, M, N, K, epi, bias.data_ptr(),
                                      gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                      out.data_ptr() if epi != 2 else None, M * N, N, 4.0, stream()))
    call()
    torch.cuda.synchronize()
    rows = torch.tensor([0, 1, 17, 255, 256, 1000, M - 129, M - 1], device=dev)
    h = (a8[rows].view(torch.float8_e4m3fn).double() @ w8.view(torch.float8_e4m3fn).double().t()) * (inv.double() / 8.0) + bias.double()
    if epi == 1:
        ref, got, tol = torch.nn.functional.gelu(h), out[rows].view(torch.float8_e4m3fn).double() / 4.0, 7e-2
    elif epi == 2:
        ref, got, tol = x0[rows].double() + gamma.double() * h, x[rows].double(), 3e-5
    else:
        ref, got, tol = h, planes_value(out)[rows], 1e-3
    err = ((got - ref).abs().max() / ref.abs().max()).item()
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
    print(f"flags={tag} fp8     {name:8s} M={M} N={N} K={K}: {med*1e3:8.1f} us  {tf:7.1f} TF algorithmic ({tf/2516.6:.3f} of bf16 peak, "
          f"{tf/5033.2:.3f} of fp8 peak)  min {ts[0]*1e3:.1f} us  err {err:.1e} {'ok' if err < tol else 'WRONG'}", flush=True)


for fmt in a.fmt:
    if fmt == "fp8":
        for name, N, K, epi in shapes:
            if not a.only or name in a.only:
                bench_fp8(name, N, K, epi)
        continue
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

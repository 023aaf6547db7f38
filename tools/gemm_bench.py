#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernel on the four encoder shapes (GPU box).  Interleaved rounds, median time, and a
correctness check against an fp64 reference on a row sample each run."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32768)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--nsplit", type=int, nargs="+", default=[1, 3])
ap.add_argument("--only", nargs="+", default=None)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
M = a.M
shapes = [("qkv", 2304, 768, 0), ("qkvK3072", 2304, 3072, 0), ("outproj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
g = torch.Generator(device="cpu").manual_seed(0)
res = {}
for ns in a.nsplit:
    for name, N, K, epi in shapes:
        if a.only and name not in a.only:
            continue
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        gamma = torch.randn(N, generator=g).to(dev)
        Ap, Wp = to_planes(A, ns), to_planes(W, ns)
        npl = Ap.shape[0]
        out = torch.zeros((npl, M, N), dtype=torch.bfloat16, device=dev) if epi != 2 else None
        x0 = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
        x = x0.clone() if epi == 2 else None

        def call():
            _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, ns, epi, bias.data_ptr(),
                                      gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                      out.data_ptr() if epi != 2 else None, M * N, N, stream()))
        call()
        torch.cuda.synchronize()
        # correctness on sampled rows
        rows = torch.tensor([0, 1, 17, 255, 256, 1000, M - 1], device=dev)
        h = planes_value(Ap)[rows] @ planes_value(Wp).t() + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(h)
        elif epi == 2:
            ref = x0[rows].double() + gamma.double() * h
        else:
            ref = h
        got = x[rows].double() if epi == 2 else planes_value(out)[rows]
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        tol = (1e-2 if ns == 1 else 1e-4) if epi != 2 else 3e-5
        ok = "ok" if err < tol else "WRONG"
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
        print(f"ns={ns} {name:8s} M={M} N={N} K={K}: {med*1e3:8.1f} us  {tf:7.1f} TF (x{ns} MFMA: {tf*ns:7.1f})  min {ts[0]*1e3:.1f} us  err {err:.1e} {ok}", flush=True)
        del A, W, Ap, Wp, out, x, x0

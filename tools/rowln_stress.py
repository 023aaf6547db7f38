#!/usr/bin/env python3
"""Randomised stress of the whole-row residual GEMM with LayerNorm in its epilogue (GPU box): for --seconds, random (M, K, format, gamma,
LayerNorm on / off) cases on random data with outlier columns; every case runs the kernel three times (four waves with private DMA rings,
counted waits, LDS reuse between main loop and epilogue: a race would show as a run-to-run difference) and compares x and both planes
BITWISE with the two launches it replaces (vtq_k_gemm epilogue 2 + vtq_k_layernorm).  Exit code 1 on any mismatch."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=90.0); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
lib = _lib.load(); dev = "cuda"; N = 768
rng = random.Random(a.seed); g = torch.Generator(device="cpu").manual_seed(a.seed)
t_end = time.time() + a.seconds
cases = bad = 0
while time.time() < t_end:
    fmt = rng.choice(["fp16x3", "fp16x3", "bf16x3"])
    K = rng.choice([128, 192, 256, 768, 768, 1024, 3072])
    tiles = rng.choice([1, 2, 3, 7, 31, 64, 200, 252, 256, 257, 300, 502]) if K <= 1024 else rng.choice([1, 2, 5, 64, 252, 260])
    M = 128 * tiles
    Mp = (M + 255) // 256 * 256
    sc = rng.choice([0.3, 1.0, 4.0])
    A = torch.randn(Mp, K, generator=g) * sc
    if rng.random() < 0.5:
        A[:, rng.randrange(K)] *= 30.0                      # an outlier channel
    W = torch.randn(N, K, generator=g) * rng.choice([0.01, 0.03, 0.2])
    A, W = A.to(dev), W.to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    use_gamma, use_ln = rng.random() < 0.7, rng.random() < 0.8
    gamma = (torch.randn(N, generator=g) + 1).to(dev) if use_gamma else None
    lw, lb = (torch.randn(N, generator=g) + 1).to(dev), torch.randn(N, generator=g).to(dev)
    x0 = (torch.randn(Mp, N, generator=g) * rng.choice([0.5, 3.0])).to(dev)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    xr = x0.clone()
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, Mp, N, K, num_code(fmt), 2, bias.data_ptr(), gamma.data_ptr() if use_gamma else None,
                              xr.data_ptr(), None, 0, 0, stream()))
    orf = torch.zeros((2, Mp, N), dtype=elt_dtype(fmt), device=dev)
    _lib.check(lib.vtq_k_layernorm(xr.data_ptr(), lw.data_ptr(), lb.data_ptr(), orf.data_ptr(), Mp * N, Mp, N, FORMATS[fmt][0], 2, stream()))
    ok = True
    for rep in range(3):
        x = x0.clone()
        out = torch.full((2, Mp, N), 3.0, dtype=elt_dtype(fmt), device=dev)
        _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(), gamma.data_ptr() if use_gamma else None,
                                        x.data_ptr(), lw.data_ptr() if use_ln else None, lb.data_ptr() if use_ln else None, out.data_ptr() if use_ln else None,
                                        Mp * N, stream()))
        torch.cuda.synchronize()
        ok &= bool(torch.equal(x[:M], xr[:M]) and torch.equal(x[M:], x0[M:]))
        if use_ln:
            ok &= bool(torch.equal(out[:, :M].view(torch.int16), orf[:, :M].view(torch.int16)))
        ok &= bool((out[:, M:] == 3.0).all()) and (use_ln or bool((out == 3.0).all()))
    cases += 1
    if not ok:
        bad += 1
        print(f"MISMATCH fmt={fmt} M={M} K={K} gamma={use_gamma} ln={use_ln}", flush=True)
    if cases % 50 == 0:
        print(f"{cases} cases, {bad} mismatches", flush=True)
print(f"rowln stress seed {a.seed}: {cases} cases x 3 runs, {bad} mismatches")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Phases of the whole-row residual GEMM from in-kernel stamps (diagnostic build: tools/build_abl.sh diag "-DVTQ_GEMM_DIAG -DVTQ_MEASURE",
VTQ_LIB_PATH=tools/_abl/diag.so): per workgroup the shader-clock cycles and 100 MHz ticks of [prologue | K loop | epilogue to the last
store's issue | store drain], after >= 1.5 s of back-to-back launches on random data."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32256); ap.add_argument("--fmt", default="fp16x3"); ap.add_argument("--warm", type=float, default=1.5)
ap.add_argument("--noln", action="store_true")
a = ap.parse_args()
lib = _lib.load(); dev = "cuda"; N = 768; M = a.M
g = torch.Generator(device="cpu").manual_seed(0)
diag = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
is_diag = lib.vtq_debug_gemm_diag(diag.data_ptr(), 0)
for name, K in (("outproj", 768), ("fc2", 3072)):
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    bias, gamma = torch.randn(N, generator=g).to(dev), (torch.randn(N, generator=g) + 1).to(dev)
    lw, lb = (torch.randn(N, generator=g) + 1).to(dev), torch.randn(N, generator=g).to(dev)
    x = torch.randn(M, N, generator=g).to(dev)
    Ap, Wp = to_planes(A, a.fmt, "a"), to_planes(W, a.fmt, "w")
    out = torch.zeros((2, M, N), dtype=torch.float16 if a.fmt.startswith("fp16") else torch.bfloat16, device=dev)

    def call():
        _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, K, num_code(a.fmt), bias.data_ptr(), gamma.data_ptr(), x.data_ptr(),
                                        None if a.noln else lw.data_ptr(), None if a.noln else lb.data_ptr(), None if a.noln else out.data_ptr(), M * N, stream()))
    call(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < a.warm:
        for _ in range(20):
            call()
        torch.cuda.synchronize(); x.normal_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    line = f"{os.environ.get('VTQ_LIB_PATH', 'shipped')} {a.fmt} {name} M={M} K={K}{' (no LN output)' if a.noln else ''}: {us:7.1f} us/launch"
    if is_diag:
        d = diag.cpu().view(256, 64)[: min(256, M // 128)]
        cyc, ticks = d[:, 0:4].double().median(0).values, d[:, 4:8].double().median(0).values
        names = ["prologue", "K loop", "epilogue", "drain"]
        parts = ", ".join(f"{n} {c / 1e3:.1f} kcyc = {t / 100:.1f} us @ {c / (t * 10 + 1e-9):.2f} GHz" for n, c, t in zip(names, cyc.tolist(), ticks.tolist()))
        nk = K // 32
        aw, ww = d[:, 8].double().median().item(), d[:, 9].double().median().item()
        line += (f" | per tile: {parts} | K loop vs MFMA floor {cyc[1].item() / (nk * 288 * 16):.3f}x | in the K loop: wait for A + barrier {aw / 1e3:.1f} kcyc "
                 f"({aw / cyc[1].item():.1%}), waits for W {ww / 1e3:.1f} kcyc ({ww / cyc[1].item():.1%}) [each incl. ~45 cycles of stamps per visit]")
    print(line, flush=True)

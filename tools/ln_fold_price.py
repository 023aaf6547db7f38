#!/usr/bin/env python3
"""Pricing of the PRODUCER side of a LayerNorm fold (GPU box; needs the pricing build: tools/build_abl.sh resid_planes
"-DVTQ_RESID_PLANES", run with VTQ_LIB_PATH=tools/_abl/resid_planes.so).

A fold would delete the stand-alone LayerNorm launches (23 per forward) and make the residual GEMMs' epilogues write, beside the
fp32 residual row, the consumer's hi / lo operand planes of that row and per-row (mean, M2) partials of their 256 columns.  The
pricing build does exactly that when vtq_k_gemm is given an output-plane pointer with the residual epilogue; this script times the
two residual GEMMs of a layer with and without it, interleaved, checks that the planes hold the new residual row and the partials
its statistics, and prints the stand-alone LayerNorm beside them.  The consumer side (two FMAs per output element and a 2 KB LDS
table per tile in the QKV / fc1 epilogues) is not priced here."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32256)
ap.add_argument("--fmt", default="fp16x3")
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
lib = _lib.load()
dev, M, fmt = "cuda", a.M, a.fmt
g = torch.Generator(device="cpu").manual_seed(0)


def timed(call, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N, K in (("outproj", 768, 768), ("fc2", 768, 3072)):
    A = to_planes(torch.randn(M, K, generator=g).to(dev), fmt, "a")
    W = to_planes((torch.randn(N, K, generator=g) * 0.03).to(dev), fmt, "w")
    bias, gamma = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    x0 = torch.randn(M, N, generator=g).to(dev)
    x = x0.clone()
    planes = torch.zeros(2, M, N, dtype=elt_dtype(fmt), device=dev)

    def call(with_planes):
        _lib.check(lib.vtq_k_gemm(A.data_ptr(), M * K, K, W.data_ptr(), N * K, M, N, K, num_code(fmt), 2, bias.data_ptr(), gamma.data_ptr(),
                                  x.data_ptr(), planes.data_ptr() if with_planes else None, M * N, N, stream()))
    x.copy_(x0); call(True); torch.cuda.synchronize()
    got = planes_value(planes)
    ok = (got - x).abs().max().item() <= 2e-6 * x.abs().max().item()          # hi + lo of the new residual row (22 significand bits)
    ts = {False: [], True: []}
    for r in range(a.rounds):
        for wp in (False, True):
            ts[wp].append(timed(lambda: call(wp)))
    for wp in ts:
        ts[wp].sort()
    t0, t1 = ts[False][len(ts[False]) // 2], ts[True][len(ts[True]) // 2]
    print(f"{name:8s} {fmt} M={M} N={N} K={K}: residual epilogue {t0:7.1f} us   + operand planes and row partials {t1:7.1f} us  ({t1 - t0:+.1f} us)"
          f"   planes = new residual row: {ok}", flush=True)

H = 768
xr = torch.randn(M, H, generator=g).to(dev)
w, b = torch.randn(H, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
out = torch.zeros(2, M, H, dtype=elt_dtype(fmt), device=dev)
ln = lambda: _lib.check(lib.vtq_k_layernorm(xr.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M * H, M, H, 1 if fmt.startswith("fp16") else 0, 2, stream()))
ln(); torch.cuda.synchronize()
tl = sorted(timed(ln) for _ in range(a.rounds))[a.rounds // 2]
print(f"stand-alone LayerNorm, M={M} H={H}: {tl:.1f} us per launch (back to back; 34.6 us inside a forward)")

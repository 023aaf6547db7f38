#!/usr/bin/env python3
"""Where the whole-row kernel differs from the two-launch path (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, stream, planes_value
lib = _lib.load(); dev = "cuda"; fmt = "fp16x3"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
N = 768
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn((max(M, 256) + 255) // 256 * 256, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
bias = torch.randn(N, generator=g).to(dev); gamma = (torch.randn(N, generator=g) + 1).to(dev) if len(sys.argv) > 3 else None; lw = (torch.randn(N, generator=g) + 1).to(dev); lb = torch.randn(N, generator=g).to(dev)
Mp = (max(M, 256) + 255) // 256 * 256
x0 = torch.randn(Mp, N, generator=g).to(dev)
Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
xa, xb = x0.clone(), x0.clone()
oa = torch.zeros((2, Mp, N), dtype=torch.float16, device=dev); ob = torch.zeros_like(oa)
_lib.check(lib.vtq_k_gemm(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, Mp, N, K, num_code(fmt), 2, bias.data_ptr(), gamma.data_ptr() if gamma is not None else None, xa.data_ptr(), None, 0, 0, stream()))
_lib.check(lib.vtq_k_layernorm(xa.data_ptr(), lw.data_ptr(), lb.data_ptr(), oa.data_ptr(), Mp * N, Mp, N, 1, 2, stream()))
_lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(), gamma.data_ptr() if gamma is not None else None, xb.data_ptr(), lw.data_ptr(), lb.data_ptr(), ob.data_ptr(), Mp * N, stream()))
torch.cuda.synchronize()
ref = x0[:M].double() + (gamma.double() if gamma is not None else 1.0) * (planes_value(Ap)[:M] @ planes_value(Wp).t() + bias.double())
d = (xa[:M] - xb[:M]).abs()
print("x: differing elements", int((d > 0).sum()), "of", d.numel(), "max abs diff", d.max().item())
print("   old vs fp64", (xa[:M].double() - ref).abs().max().item(), " new vs fp64", (xb[:M].double() - ref).abs().max().item())
bad = (d > 0)
print("   rows with a difference:", int(bad.any(1).sum()), " cols:", int(bad.any(0).sum()))
rb = bad.any(1).nonzero().flatten()[:40].tolist(); cb = bad.any(0).nonzero().flatten()[:60].tolist()
print("   first rows", rb); print("   first cols", cb)
big = (d > 1e-3)
print("   > 1e-3:", int(big.sum()), "rows", big.any(1).nonzero().flatten()[:20].tolist(), "cols", big.any(0).nonzero().flatten()[:40].tolist())
do = (oa[:, :M].float() - ob[:, :M].float()).abs()
print("planes: differing", int((do > 0).sum()), "max", do.max().item())
if gamma is not None:
    z = torch.zeros_like(x0)
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, Mp, N, K, num_code(fmt), 2, bias.data_ptr(), None, z.data_ptr(), None, 0, 0, stream()))
    torch.cuda.synchronize()
    acc = z[:M]                                    # the accumulators (bias included), exactly
    two = x0[:M] + gamma * acc                     # two roundings (torch: mul then add)
    fma = (x0[:M].double() + gamma.double() * acc.double()).float()     # one rounding
    for name, t in (("old", xa[:M]), ("new", xb[:M])):
        print(name, "== two roundings:", int((t != two).sum()), "mismatches;  == fma:", int((t != fma).sum()), "mismatches")

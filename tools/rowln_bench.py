#!/usr/bin/env python3
"""Same-box A/B of the whole-row residual GEMM with LayerNorm in its epilogue (csrc/gemm_rowln.hip) against the two launches it replaces
(256 x 256 residual GEMM + layernorm_kernel), on the encoder's out-proj (K = 768) and fc2 (K = 3072) shapes.  Interleaved rounds, HIP
events around groups of 5 launches, median; the outputs of the two paths are compared bitwise on every run."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, nargs="+", default=[32256, 64256])
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--fmt", nargs="+", default=["fp16x3"])
ap.add_argument("--tag", default=os.environ.get("VTQ_LIB_PATH", "shipped"))
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
N = 768
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for fmt in a.fmt:
    for M in a.M:
        for name, K in (("outproj", 768), ("fc2", 3072)):
            A = torch.randn(M, K, generator=g).to(dev)
            W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
            bias, gamma = torch.randn(N, generator=g).to(dev), (torch.randn(N, generator=g) + 1).to(dev)
            lw, lb = (torch.randn(N, generator=g) + 1).to(dev), torch.randn(N, generator=g).to(dev)
            x0 = torch.randn(M, N, generator=g).to(dev)
            Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
            xa, xb = x0.clone(), x0.clone()
            oa = torch.zeros((2, M, N), dtype=elt_dtype(fmt), device=dev)
            ob = torch.zeros_like(oa)

            def old_gemm():
                _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 2, bias.data_ptr(), gamma.data_ptr(),
                                          xa.data_ptr(), None, 0, 0, stream()))

            def old_ln():
                _lib.check(lib.vtq_k_layernorm(xa.data_ptr(), lw.data_ptr(), lb.data_ptr(), oa.data_ptr(), M * N, M, N, FORMATS[fmt][0], 2, stream()))

            def old():
                old_gemm(); old_ln()

            def new():
                _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(), gamma.data_ptr(),
                                                xb.data_ptr(), lw.data_ptr(), lb.data_ptr(), ob.data_ptr(), M * N, stream()))

            def new_noln():
                _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(), gamma.data_ptr(),
                                                xb.data_ptr(), None, None, None, 0, stream()))
            old(); new()
            torch.cuda.synchronize()
            same = bool(torch.equal(xa, xb) and torch.equal(oa.view(torch.int16), ob.view(torch.int16)))
            t = {"old": [], "old_gemm": [], "old_ln": [], "new": [], "new_noln": []}
            for r in range(a.rounds):
                t["old"].append(timed(old)); t["new"].append(timed(new)); t["old_gemm"].append(timed(old_gemm)); t["old_ln"].append(timed(old_ln))
                t["new_noln"].append(timed(new_noln))
            med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
            fl = 2.0 * M * N * K
            print(f"{a.tag} {fmt} {name:8s} M={M} K={K}: two launches {med['old']:7.1f} us (gemm {med['old_gemm']:.1f} + ln {med['old_ln']:.1f}) | "
                  f"whole-row {med['new']:7.1f} us ({fl / med['new'] / 1e6:6.1f} TF, {fl / med['new'] / 1e6 / 2516.6:.3f} of peak; without LN output "
                  f"{med['new_noln']:.1f}) | ratio {med['new'] / med['old']:.3f}  bitwise {'identical' if same else 'DIFFERENT'}", flush=True)

#!/usr/bin/env python3
"""Tile-shape exploration of the GEMM: every tile variant of csrc/gemm_st.hip against the persistent 256x256
kernel on the encoder's four GEMM shapes at the row counts of B = 1 .. 32 pairs (N = 500 patches): bitwise comparison of the
outputs and median time.  Variants 3+ exist in -DVTQ_GEMM_ST_EXPLORE builds only."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32])
ap.add_argument("--seq", type=int, default=501)
ap.add_argument("--variants", type=int, nargs="+", default=[0, 1, 2])
ap.add_argument("--fmt", default="fp16x3")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--only", nargs="+", default=None)
ap.add_argument("--cold", type=int, default=1, help="distinct weight buffers cycled through (1: the same W every launch, L2 / Infinity-Cache warm; 40: "
                "every launch's W comes from HBM, as every layer's does inside a forward)")
ap.add_argument("--json", default=None)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
NAMES = {-1: "rule", 0: "256x256p", 1: "64x64 r3 +2 DMA waves", 2: "64x64 r2 +4 DMA waves (2/CU)", 3: "128x128 r2 +8 DMA waves", 9: "64x64 r3", 10: "64x64 r2 (2/CU)", 18: "128x128 r2", 4: "128x64 r3", 5: "64x128 r3", 6: "128x128/4w r2", 7: "64x64 r4",
         8: "64x64 r5", 20: "64x64 r3 +4 DMA waves", 21: "64x64 r2 +4 DMA waves (2/CU)", 22: "128x128 r2 +4 DMA waves", 23: "64x64 r3 +2 DMA waves",
         24: "64x64 r4 +4 DMA waves", 25: "128x128 r2 +8 DMA waves", 11: "64x64 r3 LOADS ONLY", 12: "64x64 r3 NO DMA", 13: "64x64 r3 NO DMA NO BARRIER", 14: "64x64 r3 NO DMA NO LDS READS",
         15: "64x64 r3 MFMA ONLY", 16: "128x128 r2 LOADS ONLY", 17: "128x128 r2 NO DMA"}
shapes = [("qkv", 2304, 768, 0), ("outproj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
g = torch.Generator(device="cpu").manual_seed(0)
res = []
fmt = a.fmt
for B in a.batches:
    M = (2 * B * a.seq + 255) // 256 * 256
    for name, N, K, epi in shapes:
        if a.only and name not in a.only:
            continue
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        gamma = torch.randn(N, generator=g).to(dev)
        Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
        Wps = [Wp] + [Wp.clone() for _ in range(a.cold - 1)]
        turn = [0]
        npl = Ap.shape[0]
        x0 = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
        outs, times = {}, {}
        for v in a.variants:
            _lib.check(lib.vtq_debug_gemm_variant(v))
            out = torch.zeros((npl, M, N), dtype=elt_dtype(fmt), device=dev) if epi != 2 else None
            x = x0.clone() if epi == 2 else None

            def call():
                turn[0] = (turn[0] + 1) % a.cold
                _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wps[turn[0]].data_ptr(), N * K, M, N, K, num_code(fmt), epi, bias.data_ptr(),
                                          gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                          out.data_ptr() if epi != 2 else None, M * N, N, stream()))
            call()
            torch.cuda.synchronize()
            outs[v] = (x if epi == 2 else out).clone()
            ts = []
            for r in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            ts.sort()
            times[v] = ts[len(ts) // 2] * 1e3
        _lib.check(lib.vtq_debug_gemm_variant(-1))
        base = a.variants[0]
        line = f"B={B:2d} M={M:5d} {name:8s}"
        for v in a.variants:
            same = bool(torch.equal(outs[v].view(torch.int16 if epi != 2 else torch.int32), outs[base].view(torch.int16 if epi != 2 else torch.int32)))
            line += f" | {NAMES.get(v, v)}: {times[v]:7.1f} us{'' if same else ' DIFFERENT'}"
            res.append({"B": B, "M": M, "gemm": name, "variant": v, "us": times[v], "bitwise_equal_to_256": same})
        real = {v: t for v, t in times.items() if v < 11 or v >= 18}
        best = min(real, key=real.get)
        rule = lib.vtq_k_gemm_tile_rule(M, N, K, num_code(fmt))
        print(line + f" | best {NAMES.get(best, best)} | rule picks {NAMES.get(rule, rule)}", flush=True)
        del A, W, Ap, Wp, Wps, outs
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)

#!/usr/bin/env python3
"""Block walk of the pipelined attention kernel, A/B on one box (GPU): XCD-strided (vtq_debug_attention_map(0), the default) against the
round-3 walk (1: consecutive blocks per workgroup / paired).  Per shape: outputs compared bit for bit, sustained time per launch
(>= --warm s of back-to-back launches, then --timed launches), the two walks interleaved --rounds times."""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["64x501x768", "128x501x768", "64x521x768", "32x1025x1024", "64x1025x768", "8x2501x768", "8x5001x768", "2x5001x768"])
ap.add_argument("--fmt", default="fp16x3")
ap.add_argument("--warm", type=float, default=0.4)
ap.add_argument("--timed", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variant", type=int, default=-1)
a = ap.parse_args()
lib = _lib.load()
fmt = a.fmt
for shp in a.shapes:
    nseq, S, H = (int(v) for v in shp.split("x"))
    rows = nseq * S + 128
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(rows, 3 * H, generator=g) * 1.5).cuda()
    P = to_planes(qkv, fmt, "a")
    del qkv
    out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device="cuda")
    call = lambda: _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
    _lib.check(lib.vtq_debug_attention_variant(a.variant))
    res, outs = {0: [], 1: []}, {}
    for rnd in range(a.rounds):
        for m in (1, 0):
            _lib.check(lib.vtq_debug_attention_map(m))
            out.zero_()
            call(); torch.cuda.synchronize()
            if rnd == 0:
                outs[m] = out[:, : nseq * S].clone()
            t0 = time.time()
            while time.time() - t0 < a.warm:
                for _ in range(10):
                    call()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.timed):
                call()
            e1.record(); torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / a.timed * 1e3)
    _lib.check(lib.vtq_debug_attention_map(0)); _lib.check(lib.vtq_debug_attention_variant(-1))
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    fl = 4.0 * nseq * (H // 64) * S * S * 64
    t1, t0_ = statistics.median(res[1]), statistics.median(res[0])
    print(f"{shp:14s} {fmt}: round-3 walk {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TF)   XCD-strided {t0_:8.1f} us ({fl / t0_ / 1e6:6.1f} TF)  {t0_ / t1 - 1:+.1%}   "
          f"bit-identical: {same}   rounds: {['%.1f' % v for v in res[1]]} / {['%.1f' % v for v in res[0]]}", flush=True)
    del P, out, outs
    torch.cuda.empty_cache()

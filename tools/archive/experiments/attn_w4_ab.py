#!/usr/bin/env python3
"""The one-wave-per-SIMD attention kernel (variant 3: 4 waves x 2 row blocks, csrc/attention_w4.hip) against the 8-wave pipelined kernel (variant 1),
GPU box: outputs bit for bit, sustained time per launch, interleaved."""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["64x501x768", "64x512x768", "32x1024x1024", "8x2501x768", "8x5001x768", "12x300x768", "5x575x768"])
ap.add_argument("--fmt", nargs="+", default=["fp16x3", "bf16x3"])
ap.add_argument("--warm", type=float, default=0.4)
ap.add_argument("--timed", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
lib = _lib.load()
for shp in a.shapes:
    nseq, S, H = (int(v) for v in shp.split("x"))
    rows = nseq * S + 128
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(rows, 3 * H, generator=g) * 1.5).cuda()
    qkv[S - 3, H:H + 64] *= 6.0
    for fmt in a.fmt:
        P = to_planes(qkv, fmt, "a")
        out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device="cuda")
        call = lambda: _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
        res, outs = {1: [], 3: []}, {}
        for rnd in range(a.rounds):
            for v in (1, 3):
                _lib.check(lib.vtq_debug_attention_variant(v))
                out.fill_(7.0)
                call(); torch.cuda.synchronize()
                if rnd == 0:
                    outs[v] = out[:, : nseq * S].clone()
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
                res[v].append(e0.elapsed_time(e1) / a.timed * 1e3)
        _lib.check(lib.vtq_debug_attention_variant(-1))
        same = torch.equal(outs[1].view(torch.int16), outs[3].view(torch.int16))
        nbad = 0 if same else int((outs[1].view(torch.int16) != outs[3].view(torch.int16)).sum().item())
        nh = H // 64
        x = planes_value(P)[: 2 * S].view(2, S, 3, nh, 64)
        q, k, v_ = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v_).permute(0, 2, 1, 3).reshape(2, S, H)
        err = ((planes_value(outs[3])[: 2 * S].view(2, S, H) - ref).abs().max() / ref.abs().max()).item()
        fl = 4.0 * nseq * nh * S * S * 64
        t1, t3 = statistics.median(res[1]), statistics.median(res[3])
        print(f"{shp:14s} {fmt:7s}: 8-wave {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TF)   4-wave x 2 blocks {t3:8.1f} us ({fl / t3 / 1e6:6.1f} TF)  {t3 / t1 - 1:+.1%}   "
              f"bit-identical: {same}{'' if same else f' ({nbad} differ)'}  err vs fp64 {err:.1e}", flush=True)
        del P, out, outs
    del qkv
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""The two attention kernels (4-wave, software-pipelined) side by side (GPU box): outputs compared bit for bit, error against an fp64
reference, time per launch (interleaved short bursts; tools/attn_probe.py times sustained launches)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["64x501x768", "32x1025x1024", "3x51x768", "5x521x768", "2x64x768", "8x2501x768"])
ap.add_argument("--fmt", nargs="+", default=["fp16x3", "fp16", "bf16x3", "bf16"])
ap.add_argument("--reps", type=int, default=20)
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
        outs, times = [], []
        for variant in (0, 1):
            lib.vtq_debug_attention_variant(variant)
            out = torch.full((P.shape[0], rows, H), 7.0, dtype=elt_dtype(fmt), device="cuda")
            call = lambda: _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
            call(); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    call()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.reps * 1e3)
            ts.sort()
            outs.append(out[:, : nseq * S].clone()); times.append(ts[2])
        lib.vtq_debug_attention_variant(-1)
        same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        nh = H // 64
        x = planes_value(P)[: 2 * S].view(2, S, 3, nh, 64)
        q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(2, S, H)
        errs = [((planes_value(o)[: 2 * S].view(2, S, H) - ref).abs().max() / ref.abs().max()).item() for o in outs]
        nbad = 0 if same else int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum().item())
        print(f"{shp:14s} {fmt:7s}: 4-wave {times[0]:7.1f} us  pipelined {times[1]:7.1f} us ({times[1] / times[0] - 1:+.1%})  bit-identical: {same}"
              f"{'' if same else f' ({nbad} elements differ)'}  err {errs[0]:.1e} / {errs[1]:.1e}", flush=True)

#!/usr/bin/env python3
"""Soak run (GPU box): the bench workload (B = 32, N = 500) N times per mode, every result compared bit for bit with the first --
a race / uninitialised-memory detector at full size (126 x 12 tiles per GEMM launch, all CUs, chained tile lists) -- and timed:
the rate of the first 20 forwards against the rate of the whole run (does the number hold once the chip is warm?)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, synth
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=200); ap.add_argument("--modes", nargs="+", default=["fp16x3", "fp16x2", "fp16", "bf16x3", "fp8"])
a = ap.parse_args()
bad = 0
for prec in a.modes:
    m = model_class(prec)(vit_config=dict(variant="ViT-B16", pretrained=False), precision=prec)
    sd = synth.make_state_dict(m.spec, 0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    patches, pos, _ = synth.make_inputs(m.spec, 32, 500, 7)
    tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
    args = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))
    with torch.no_grad():
        q0 = m(*args)[0].clone()
        torch.cuda.synchronize()
        qs = []
        t0 = time.perf_counter()
        t20 = None
        for i in range(a.reps):
            qs.append(m(*args)[0])
            if i == 19:
                torch.cuda.synchronize()
                t20 = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        diff = sum(0 if torch.equal(q, q0) else 1 for q in qs)
    print(f"{prec}: {a.reps} repeated forwards at B=32, N=500: {diff} differ from the first; {32 * a.reps / dt:.0f} pairs/s over {dt:.2f} s "
          f"({dt / a.reps * 1e3:.2f} ms per forward; the first 20: {32 * 20 / t20:.0f} pairs/s)", flush=True)
    bad += diff
    del m; torch.cuda.empty_cache()
sys.exit(1 if bad else 0)

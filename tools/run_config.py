#!/usr/bin/env python3
"""Run one BASELINE config shape end to end on the GPU: timing (both numerics modes) + parity of a few pairs vs the oracle."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vtamiq_amd import VTAMIQ, synth
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise
from oracle import vtamiq_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="ViT-L16"); ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--patches", type=int, default=1024); ap.add_argument("--scales", type=int, default=3)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--refdefault", action="store_true", help="the reference's default topology (train_config.py:169-194): 6 layers, 8 register tokens, LayerScale, r = 16")
a = ap.parse_args()
kw = dict(vit_config=dict(variant=a.variant, num_scales=a.scales, pretrained=False))
if a.refdefault:
    kw = dict(vit_config=dict(variant=a.variant, num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True, num_scales=a.scales, pretrained=False), ca_reduction=16)
from vtamiq_amd import _lib
for prec in ("fp16x3", "fp16x2", "fp16", "bf16") + (("fp8",) if _lib.fp8_available() else ()):      # fp8: the experiment's own library
    m = model_class(prec)(**json.loads(json.dumps(kw)), precision=prec)
    spec = m.spec
    sd = synth.make_state_dict(spec, 0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    patches, pos, scales = synth.make_inputs(spec, a.batch, a.patches, 7)
    tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
    ts = torch.from_numpy(scales).float().cuda() if scales is not None else None
    args = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()),
            (ts[:, 0].contiguous(), ts[:, 1].contiguous()) if ts is not None else (None, None))
    with torch.no_grad():
        for _ in range(2): q = m(*args)[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): q = m(*args)[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    fl = spec.flops_per_pair_executed(a.patches, cls_prune=(prec != "fp8"))      # fp8 runs the full last layer
    print(f"{a.variant} B={a.batch} N={a.patches} scales={a.scales} {prec}: {dt*1e3:.2f} ms/step  {a.batch/dt:.1f} pairs/s  "
          f"{a.batch/dt*fl/2.5166e15*100:.1f}% of bf16 MFMA roofline (executed flops); workspace {m.workspace_bytes(a.batch, a.patches)/2**30:.2f} GiB", flush=True)
    if a.check:
        n = a.check
        cin = ((tp[:n, 0].cpu(), tp[:n, 1].cpu()), (tq[:n, 0].cpu(), tq[:n, 1].cpu()),
               (ts[:n, 0].cpu(), ts[:n, 1].cpu()) if ts is not None else (None, None))
        torch.set_num_threads(16)
        qr = O.vtamiq_forward(O.to_torch(sd), spec, *cin)[0].numpy()
        d = np.abs(q[:n].cpu().numpy() - qr)
        print(f"   parity vs oracle ({n} pair): q_ref={qr} max_rel={float((d/np.abs(qr)).max()):.2e}", flush=True)
    del m; torch.cuda.empty_cache()

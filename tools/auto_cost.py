#!/usr/bin/env python3
"""What the drop-in's default precision="auto" costs against an explicit "fp16x3" (GPU box): the same engine mode, plus the error
word read (4-byte copy + stream synchronisation) after every forward.  BASELINE configs[1] shape, 40 forwards each, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, synth

dev = torch.device("cuda")
B, N = 32, 500
g = torch.Generator(device="cpu").manual_seed(0)
pr, pd = (torch.randn(B, N, 3, 16, 16, generator=g).to(dev) for _ in range(2))
qr, qd = (torch.rand(B, N, 2, generator=g).to(dev) * 0.999 for _ in range(2))
models = {}
for prec in ("fp16x3", "auto"):
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", pretrained=False), precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 0).items()})
    models[prec] = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            models[prec]((pr, pd), (qr, qd), (None, None))
torch.cuda.synchronize()
res = {k: [] for k in models}
with torch.no_grad():
    for rep in range(3):
        for prec, m in models.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40):
                q, _ = m((pr, pd), (qr, qd), (None, None))
            torch.cuda.synchronize()
            res[prec].append((time.perf_counter() - t0) / 40 * 1e3)
for prec, ts in res.items():
    print(f"precision={prec!r}: {min(ts):.3f} ms per forward (best of 3 x 40), {B / min(ts) * 1e3:.1f} pairs/s")
print(f"auto / fp16x3 = {min(res['auto']) / min(res['fp16x3']):.4f}")

#!/usr/bin/env python3
"""Measure the on-device image -> patch path and the pairwise entry point (8f-2) on the GPU box."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vtamiq_amd import VTAMIQ, synth
from vtamiq_amd.patches import extract_patches

dev = "cuda"
# ---- 8f-1: 64 images (32 pairs) of 384x512, 500 patches each, single scale and 3 scales
NI, H, W, N = 64, 384, 512, 500
g = torch.Generator(device=dev).manual_seed(0)
imgs = torch.randint(0, 256, (NI, H, W, 3), device=dev, dtype=torch.uint8, generator=g)
for nsc in (1, 3):
    counts = synth.num_patches_per_scale(N, nsc) if nsc > 1 else np.array([N])
    sid = np.concatenate([np.full(c, s) for s, c in enumerate(counts)]).astype(np.int32)
    smp = np.zeros((NI, N, 2), np.int32)
    rs = np.random.RandomState(0)
    for s in range(nsc):
        h, w = H >> s, W >> s
        m = sid == s
        smp[:, m, 0] = rs.randint(0, h - 15, size=(NI, m.sum())); smp[:, m, 1] = rs.randint(0, w - 15, size=(NI, m.sum()))
    smp_d, sid_d = torch.from_numpy(smp).to(dev), torch.from_numpy(np.broadcast_to(sid, (NI, N)).copy()).to(dev)
    for _ in range(3): out = extract_patches(imgs, smp_d, sid_d if nsc > 1 else None, nsc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = extract_patches(imgs, smp_d, sid_d if nsc > 1 else None, nsc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    byts = NI * H * W * 3 + NI * 3 * H * W * 4 * 2 + NI * N * 768 * 4 * 2
    print(f"8f-1 extract_patches: {NI} images {H}x{W}, {N} patches, {nsc} scale(s): {dt*1e3:.3f} ms / batch = {NI/2/dt:.0f} pairs/s "
          f"(~{byts/dt/1e12:.2f} TB/s of algorithmic traffic)")
# ---- 8f-2: pairwise triplets, fused vs two calls
m = VTAMIQ(vit_config=dict(variant="ViT-B16", pretrained=False), precision=os.environ.get("VTAMIQ_PRECISION", "fp16x3"))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 0).items()}); m = m.to(dev).eval()
B = 32
p = [torch.rand(B, N, 3, 16, 16, device=dev) * 2 - 1 for _ in range(3)]
q = [torch.rand(B, N, 2, device=dev).clamp_(max=1 - 1e-6) for _ in range(3)]
def two():
    return m((p[0], p[1]), (q[0], q[1]), (None, None))[0], m((p[0], p[2]), (q[0], q[2]), (None, None))[0]
def fused():
    return m.forward_pairwise(p, q, None)
with torch.no_grad():
    for fn, name in ((two, "two model calls (reference structure)"), (fused, "forward_pairwise (ref encoded once)")):
        for _ in range(2): r = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): r = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        print(f"8f-2 {name}: {dt*1e3:.2f} ms / {B} triplets = {B/dt:.0f} triplets/s [{m.precision}]")
    a, b = two(); c, d = fused()
    print("bit-identical:", bool(torch.equal(a, c) and torch.equal(b, d)))

#!/usr/bin/env python3
"""Stage times of the end-to-end validation loop (bench.py e2e_validation): H2D rate of pinned buffers, host sampling, gather, forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device("cuda", 0)
for mb in (1, 8, 38, 151):
    h = torch.empty(mb * 1024 * 1024, dtype=torch.uint8).pin_memory()
    d = torch.empty_like(h, device=dev)
    for _ in range(2): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    hp = torch.empty(mb * 1024 * 1024, dtype=torch.uint8)
    t0 = time.perf_counter(); d.copy_(hp); torch.cuda.synchronize(); dtp = time.perf_counter() - t0
    print(f"H2D {mb:4d} MiB: pinned {mb / 1024 / dt:6.2f} GiB/s ({dt * 1e3:.2f} ms), pageable {mb / 1024 / dtp:6.2f} GiB/s")
rs = np.random.RandomState(0)
t0 = time.perf_counter()
for _ in range(10):
    half = np.stack([rs.randint(0, 369, size=(32, 500)), rs.randint(0, 497, size=(32, 500))], axis=-1).astype(np.int32)
print(f"host sampling of one batch: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
from vtamiq_amd.patches import extract_patches, check_samples_host
img = torch.randint(0, 256, (64, 384, 512, 3), dtype=torch.uint8, device=dev)
smp = torch.from_numpy(np.concatenate([half, half])).to(dev)
for _ in range(3): extract_patches(img, smp, validate=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): extract_patches(img, smp, validate=False)
torch.cuda.synchronize(); print(f"extract_patches (no validation): {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms / batch")
hs = torch.from_numpy(np.concatenate([half, half])).pin_memory()
t0 = time.perf_counter()
for _ in range(10): check_samples_host(hs, None, 384, 512)
print(f"check_samples_host: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")

# ---- the loop by elimination ----------------------------------------------------------------------------------------
from vtamiq_amd import VTAMIQ, synth
B, N, H, W = 32, 500, 384, 512
m = VTAMIQ(vit_config=dict(variant="ViT-B16", pretrained=False), precision="fp16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 0).items()}); m = m.to(dev).eval()
NI = 2 * B
host_img = [torch.from_numpy(rs.randint(0, 256, size=(NI, H, W, 3), dtype=np.uint8)).pin_memory() for _ in range(2)]
host_smp = [torch.from_numpy(np.concatenate([half, half])).pin_memory() for _ in range(2)]
dev_img = [torch.empty(NI, H, W, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
dev_smp = [torch.empty(NI, N, 2, dtype=torch.int32, device=dev) for _ in range(2)]
cs = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream(dev)
copied = [torch.cuda.Event() for _ in range(2)]; consumed = [torch.cuda.Event() for _ in range(2)]
fixed = extract_patches(dev_img[0].random_(0, 256), dev_smp[0].copy_(host_smp[0]), validate=False)
fixed = (fixed[0].clone(), fixed[1].clone())

def loop(do_sample, do_check, do_upload, do_extract, n=16):
    for ev in consumed: ev.record(main)
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            s = (i + 1) % 2
            if do_sample:
                consumed[s].synchronize()
                a = np.stack([rs.randint(0, H - 15, size=(B, N)), rs.randint(0, W - 15, size=(B, N))], axis=-1).astype(np.int32)
                host_smp[s].numpy()[:B] = a; host_smp[s].numpy()[B:] = a
            if do_check: check_samples_host(host_smp[s], None, H, W)
            if do_upload:
                with torch.cuda.stream(cs):
                    cs.wait_event(consumed[s]); dev_img[s].copy_(host_img[s], non_blocking=True); dev_smp[s].copy_(host_smp[s], non_blocking=True); copied[s].record(cs)
            s = i % 2
            if do_upload: main.wait_event(copied[s])
            if do_extract: pa, po, _ = extract_patches(dev_img[s], dev_smp[s], validate=False)
            else: pa, po = fixed
            consumed[s].record(main)
            q = m((pa[:B], pa[B:]), (po[:B], po[B:]), (None, None))[0]
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for flags in ((0, 0, 0, 0), (0, 0, 0, 1), (0, 0, 1, 1), (1, 0, 1, 1), (1, 1, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0)):
    loop(*flags, n=4)
    print(f"sample {flags[0]} check {flags[1]} upload {flags[2]} extract {flags[3]}: {loop(*flags):7.2f} ms / batch")

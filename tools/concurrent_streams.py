#!/usr/bin/env python3
"""Single-query serving: K model instances (K engines, K streams) each scoring B pairs per forward, concurrently on one GPU -- against one instance and against
batching the same pairs into one forward.  A B = 1 forward is a chain of 133 short dependent launches that leaves most of the chip idle; independent chains
on separate streams can fill it."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, synth

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4]); ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--patches", type=int, default=500); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--precision", default="fp16x3")
a = ap.parse_args()
kw = dict(vit_config=dict(variant="ViT-B16", pretrained=False))
spec = VTAMIQ(**kw, precision=a.precision).spec
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec, 0).items()}
K = max(a.streams)
models, streams, inputs = [], [], []
for i in range(K):
    m = VTAMIQ(**kw, precision=a.precision); m.load_state_dict(sd); models.append(m.cuda().eval())
    streams.append(torch.cuda.Stream())
    patches, pos, _ = synth.make_inputs(spec, a.batch, a.patches, 100 + i)
    tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
    inputs.append(((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None)))
torch.cuda.synchronize()
print(f"# {a.precision} B={a.batch} pairs per forward, N={a.patches}, ViT-B/16 L=12; K instances on K streams, {a.steps} forwards each")
base = None
with torch.no_grad():
    for k in a.streams:
        for rep in range(2):                      # first repetition warms up (engine creation, workspace)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                for i in range(k):
                    with torch.cuda.stream(streams[i]):
                        q = models[i](*inputs[i])[0]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rate = k * a.batch * a.steps / dt
        base = base or rate
        print(f"K={k}: {dt / a.steps * 1e3:7.3f} ms per round of {k} forwards  {rate:8.1f} pairs/s  ({rate / base:.2f}x one instance)")
    # the same pairs batched into ONE forward of one instance
    for k in a.streams[1:]:
        patches, pos, _ = synth.make_inputs(spec, a.batch * k, a.patches, 7)
        tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
        arg = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps): models[0](*arg)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"batched B={a.batch * k}: {dt / a.steps * 1e3:7.3f} ms per forward  {a.batch * k * a.steps / dt:8.1f} pairs/s")

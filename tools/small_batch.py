#!/usr/bin/env python3
"""Small-batch / single-query regime: ms per forward and per-kernel-class time at B in {1, 2, 4, 8, 16, 32}, N = 500,
ViT-B/16 L = 12 (and the reference-default topology at B in {1, 16}).  The reference's own FLOP probe is batch 1 x 500 patches
(modules/utils.py:68-78)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, synth, _lib

PEAK = 2516.6e12
ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32])
ap.add_argument("--patches", type=int, default=500)
ap.add_argument("--precision", default="fp16x3")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--refdefault", action="store_true")
ap.add_argument("--classes", action="store_true", help="per-kernel-class table (HIP events around every launch class)")
ap.add_argument("--json", default=None)
a = ap.parse_args()
kw = (dict(vit_config=dict(variant="ViT-B16", num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True, pretrained=False), ca_reduction=16)
      if a.refdefault else dict(vit_config=dict(variant="ViT-B16", pretrained=False)))
m = VTAMIQ(precision=a.precision, **kw)
sd = synth.make_state_dict(m.spec, 0)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
rows = []
print(f"# {a.precision}  N={a.patches}  {'reference-default topology (L=6, T=9)' if a.refdefault else 'ViT-B/16 L=12 T=1'}  steps={a.steps}")
print(f"{'B':>3s} {'rows':>6s} {'ms/forward':>11s} {'ms/pair':>9s} {'pairs/s':>9s} {'mfma_frac':>9s}")
for B in a.batches:
    patches, pos, _ = synth.make_inputs(m.spec, B, a.patches, 7)
    tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
    args = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))
    with torch.no_grad():
        for _ in range(5): m(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        th = time.perf_counter()
        for _ in range(a.steps): m(*args)
        host_ms = (time.perf_counter() - th) / a.steps * 1e3       # host time to ENQUEUE a forward (the queue is deep enough not to block at these step counts)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        # one forward at a time (latency of a single query: the host waits for every score)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(*args); torch.cuda.synchronize()
        ms_sync = (time.perf_counter() - t0) / a.steps * 1e3
        f = m.spec.flops_per_pair_executed(a.patches, cls_prune=True)
        row = {"B": B, "rows": 2 * B * m.spec.seq_len(a.patches), "ms_per_forward": ms, "ms_per_forward_synchronous": ms_sync, "host_enqueue_ms_per_forward": host_ms, "ms_per_pair": ms / B,
               "pairs_per_s": B / ms * 1e3, "forward_mfma_frac": B / ms * 1e3 * f / PEAK}
        print(f"{B:3d} {row['rows']:6d} {ms:11.3f} {ms / B:9.3f} {row['pairs_per_s']:9.1f} {row['forward_mfma_frac']:9.4f}   (synchronous: {ms_sync:.3f} ms; host enqueue {host_ms:.3f} ms)")
        if a.classes:
            m.profile_enable(list(_lib.KERNEL_CLASSES))
            for _ in range(a.steps): m(*args)
            prof = m.profile_collect(); m.profile_enable([])
            row["classes"] = {}
            for name in _lib.KERNEL_CLASSES:
                s_ms, n = prof[name]
                if n:
                    row["classes"][name] = {"us_per_launch": s_ms / n * 1e3, "launches_per_step": n // a.steps, "ms_per_step": s_ms / a.steps}
                    print(f"      {name:10s} {s_ms / a.steps:8.3f} ms/step {n // a.steps:3d} launches {s_ms / n * 1e3:8.1f} us/launch")
    rows.append(row)
if a.json:
    json.dump({"precision": a.precision, "patches": a.patches, "refdefault": a.refdefault, "rows": rows}, open(a.json, "w"), indent=1)

#!/usr/bin/env python3
"""Per-kernel-class time of one forward (HIP events on the launch stream) for the bench workload."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import VTAMIQ, synth, _lib
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--patches", type=int, default=500)
ap.add_argument("--refdefault", action="store_true")
ap.add_argument("--fused-ln", action="store_true", help="vtq_config.options & VTQ_OPT_FUSED_LAYERNORM: LayerNorm inside the residual GEMMs (csrc/gemm_rowln.hip)")
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--precision", nargs="+", default=["fp16x3", "fp16x2", "fp16", "fp8"])
a = ap.parse_args()
for prec in a.precision:
    m = model_class(prec)(precision=prec, engine_options=_lib.OPT_FUSED_LAYERNORM if a.fused_ln else 0, **(dict(vit_config=dict(variant="ViT-B16", num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True, pretrained=False),
                                       ca_reduction=16) if a.refdefault else dict(vit_config=dict(variant="ViT-B16", pretrained=False))))
    sd = synth.make_state_dict(m.spec, 0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    patches, pos, _ = synth.make_inputs(m.spec, a.batch, a.patches, 7)
    tp, tq = torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda()
    args = ((tp[:, 0].contiguous(), tp[:, 1].contiguous()), (tq[:, 0].contiguous(), tq[:, 1].contiguous()), (None, None))
    with torch.no_grad():
        for _ in range(3): m(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps): m(*args)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        m.profile_enable(list(_lib.KERNEL_CLASSES))
        for _ in range(a.steps): m(*args)
        prof = m.profile_collect()
    ms = [prof[k][0] for k in _lib.KERNEL_CLASSES]; n = [prof[k][1] for k in _lib.KERNEL_CLASSES]
    tot = sum(ms) / a.steps
    print(f"{prec} B={a.batch} N={a.patches}{' LayerNorm inside the residual GEMMs' if a.fused_ln else ''}: {dt*1e3:.2f} ms/step unprofiled; classes sum {tot:.2f} ms")
    for k, name in enumerate(_lib.KERNEL_CLASSES):
        if n[k]:
            print(f"   {name:12s} {ms[k]/a.steps:7.3f} ms/step  {n[k]//a.steps:3d} launches  {ms[k]/n[k]*1e3:8.1f} us/launch")
    del m; torch.cuda.empty_cache()

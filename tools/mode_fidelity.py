#!/usr/bin/env python3
"""What the numerics modes cost in the reference's own metric (GPU box).

256 synthetic pairs on a distortion ladder (sigma in bench.LADDER_SIGMAS of bench.synth_inputs_on_device's noise), ViT-B/16 L=12,
N=500 patches; target = the fp32 oracle's scores on the host, prediction = each mode's scores; SROCC / KROCC / PLCC / RMSE through
vtamiq_amd.validate.compute_correlations (HIP rank / Kendall / Pearson kernels + the reference's logistic fit,
utils/misc/correlations.py:21-51).  Two weight sets: the flat seeded init and tests.helpers.stress_state(qk=5) (trained-like
statistics).  fp8 --calib: the fp8 mode with activation scales calibrated on the first ladder chunk (vtq_calibrate_fp8).

    python3 tools/mode_fidelity.py [--pairs 256] [--weights flat stress5]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tests.helpers import stress_state
from vtamiq_amd import VTAMIQ, synth
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--patches", type=int, default=500)
ap.add_argument("--weights", nargs="+", default=["flat", "stress5"])
ap.add_argument("--modes", nargs="+", default=["fp16x3", "bf16x3", "fp16x2", "fp16", "bf16", "fp8", "fp8-static"])
a = ap.parse_args()
dev = torch.device("cuda")
kw = dict(vit_config=dict(variant="ViT-B16", pretrained=False))
spec = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp16x3").spec
print(f"# {a.pairs} pairs, N={a.patches}, ladder sigmas {bench.LADDER_SIGMAS}; target = fp32 oracle (host, {bench.effective_cores()} cores)")
print("# columns: SROCC KROCC PLCC RMSE (logistic fit, scores min-max normalised as in the reference) | PLCC RMSE without the fit | worst raw relative error over scores >= 0.1 rms")
for wname in a.weights:
    sd_np = synth.make_state_dict(spec, 0) if wname == "flat" else stress_state(spec, 0, qk=float(wname.replace("stress", "")))
    state = {k: torch.from_numpy(v) for k, v in sd_np.items()}

    def make_model(prec):
        static = prec == "fp8-static"                       # the round-2 constants instead of the calibration on the first batch
        from vtamiq_amd import _lib
        m = model_class("fp8" if static else prec)(**json.loads(json.dumps(kw)), precision="fp8" if static else prec, engine_options=_lib.OPT_FP8_STATIC_SCALES if static else 0)
        m.load_state_dict(state)
        return m.to(dev).eval()
    r = bench.mode_fidelity(torch, make_model, spec, sd_np, a.modes, dev, pairs=a.pairs, N=a.patches, threads=min(bench.effective_cores(), 64))
    print(f"weights = {wname}: rms(q_ref) = {r['rms_q_ref']:.4e}   (oracle: {r['oracle_seconds']:.0f} s)")
    for prec, c in r["modes"].items():
        print(f"    {prec:10s} {c['SROCC']:.6f} {c['KROCC']:.6f} {c['PLCC']:.6f} {c['RMSE']:.3e} | {c['PLCC_NOFIT']:.6f} {c['RMSE_NOFIT']:.3e} | {c['max_rel_err_big_scores']:.2e}", flush=True)

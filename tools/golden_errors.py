#!/usr/bin/env python3
"""Prints the HIP-vs-golden error of every end-to-end golden case in every numerics mode (GPU box): the RAW per-score relative
error (max over all scores, and max over the scores with |q_ref| >= 0.1 rms), the rms-normalised error, and the gate's value."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import E2E_CASES, STRESS_CASES, gate_error, load_case, rel_err, split_inputs
from vtamiq_amd import VTAMIQ
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise
MODES = ("fp16x3", "fp16x2", "bf16x3", "fp16", "bf16", "fp8")       # fp8: a different model (oracle/fp8_oracle.py), distance reported only
print("# |q - q_ref| / |q_ref| against the goldens captured from the imported reference (fp32 CPU); min|q_ref|/rms shows how close to")
print("# zero the smallest score of the case is.  columns per mode: raw max over all scores | raw max over |q_ref| >= 0.1 rms | max |d| / rms | gate")
worst = {m: 0.0 for m in MODES}
for name in E2E_CASES + STRESS_CASES:      # the last two: the reference on trained-like statistics (stress_state, qk = 3 / 5)
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, s3 = split_inputs(patches, pos, scales, device="cuda")
    ref = g["q"].astype(np.float64)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    print(f"{name}: B={len(ref)} rms(q_ref)={rms:.3e} min|q_ref|/rms={np.abs(ref).min() / rms:.3f}"
          + (f"   (reference fp32 vs its own float64 scores: {np.max(np.abs(ref - g['q64']) / np.abs(g['q64'])):.2e})" if "q64" in g else ""))
    for prec in MODES:
        if prec == "fp8" and spec.num_adapters > 0:
            print("    fp8     (adapters are not available in the fp8 mode)")
            continue
        m = model_class(prec)(**json.loads(json.dumps(kw)), precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
        with torch.no_grad():
            q = m(p, ps, s3)[0].cpu().numpy().astype(np.float64)
        d = np.abs(q - ref)
        big = np.abs(ref) >= 0.1 * rms
        ge = gate_error(q, ref)
        worst[prec] = max(worst[prec], ge)
        print(f"    {prec:7s} raw_all {np.max(d / np.abs(ref)):.2e} | raw_big {np.max(d[big] / np.abs(ref[big])):.2e} | rms_norm {d.max() / rms:.2e} | gate {ge:.2e}", flush=True)
        del m
print("worst gate value per mode:", {k: f"{v:.2e}" for k, v in worst.items()})

#!/usr/bin/env python3
"""Prints the HIP-vs-golden error of every end-to-end golden case in both numerics modes (GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import E2E_CASES, load_case, rel_err, split_inputs
from vtamiq_amd import VTAMIQ
for name in E2E_CASES:
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, s3 = split_inputs(patches, pos, scales, device="cuda")
    line = f"{name:20s}"
    for prec in ("bf16x3", "bf16"):
        m = VTAMIQ(**json.loads(json.dumps(kw)), precision=prec)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
        with torch.no_grad():
            q = m(p, ps, s3)[0].cpu().numpy()
        e = rel_err(q, g["q"])
        line += f"  {prec}: max_rel {e['max_rel']:.2e} rms-normalised {e['max_rel_rms']:.2e}"
    print(line, flush=True)

"""Drift fixture for oracle/fp8_oracle.py: the fake-quant fp8 model's scores (float64 arithmetic between the rounding points)
on two of the seeded golden cases.  These are this repo's own definition of the fp8 mode (the reference has no fp8 path);
the fixture only guards the definition against accidental change.   python tests/golden/make_fp8_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fp8_oracle as F8  # noqa: E402
from oracle import vtamiq_oracle as O  # noqa: E402
from tests.helpers import GOLDEN, load_case, split_inputs  # noqa: E402

out = {}
for name in ["c1_b2_n50", "scales3_b2_n40"]:
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, dtype=torch.float64)
    q = F8.vtamiq_forward(O.to_torch(sd, torch.float64), spec, p, ps, sc)[0].numpy()
    out[name] = q
    print(name, q, "fp32 model:", g["q"])
np.savez(os.path.join(GOLDEN, "fp8_model.npz"), **out)

#!/usr/bin/env python3
"""Randomised stress of the whole forward (GPU box): bitwise batch invariance and run-to-run determinism in EVERY numerics mode on
trained-like weights (tests/helpers.stress_state with a random query / key gain, so that the peaked-softmax paths of the attention
kernels are taken).  For --seconds: a random model (ViT-B/16 trunk of 2 .. 4 layers, random head topology), random B and N;
the scores of the full batch are compared bit for bit with (i) a second run, (ii) the first k pairs run alone, (iii) the batch in
reversed order.  Exit code 1 on any mismatch."""
import argparse, json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vtamiq_amd import VTAMIQ, synth
from tests.helpers import stress_state, split_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=150.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
dev = torch.device("cuda")
t_end = time.time() + a.seconds
cases = bad = 0
while time.time() < t_end:
    L = rng.choice([2, 3, 4])
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=L, pretrained=False), num_rgs=rng.choice([1, 2, 4]), num_rcabs=rng.choice([1, 2, 4]))
    spec = VTAMIQ(**json.loads(json.dumps(kw))).spec
    sd = stress_state(spec, rng.randrange(1 << 20), qk=rng.choice([1.0, 3.0, 5.0, 8.0]))
    B = rng.choice([2, 3, 5, 8, 13])
    N = rng.choice([9, 40, 63, 64, 90, 127, 200, 299, 500])
    patches, pos, sc = synth.make_inputs(spec, B, N, rng.randrange(1 << 20))
    p, ps, s3 = split_inputs(patches, pos, sc, device=dev)
    k = rng.randint(1, B - 1)
    sub = lambda t: tuple(None if x is None else x[:k].contiguous() for x in t)
    rev = lambda t: tuple(None if x is None else x.flip(0).contiguous() for x in t)
    for prec in ("fp16x3", "fp16x2", "fp16", "bf16x3", "bf16"):
        m = VTAMIQ(**json.loads(json.dumps(kw)), precision=prec)
        m.load_state_dict({n: torch.from_numpy(np.ascontiguousarray(v)) for n, v in sd.items()})
        m = m.to(dev).eval()
        with torch.no_grad():
            q = m(p, ps, s3)[0].clone()
            q2 = m(p, ps, s3)[0].clone()
            qs = m(sub(p), sub(ps), sub(s3))[0].clone()
            qr = m(rev(p), rev(ps), rev(s3))[0].clone()
        ok = bool(torch.isfinite(q).all()) and torch.equal(q, q2) and torch.equal(q[:k], qs) and torch.equal(q, qr.flip(0))
        cases += 1
        if not ok:
            bad += 1
            print(f"MISMATCH {prec} L={L} B={B} N={N} k={k}: finite {bool(torch.isfinite(q).all())}, repeat {torch.equal(q, q2)}, "
                  f"subset {torch.equal(q[:k], qs)}, reversed {torch.equal(q, qr.flip(0))}", flush=True)
        del m
    if cases % 50 == 0:
        print(f"{cases} cases, {bad} bad", flush=True)
print(f"forward stress: {cases} (model, batch, mode) cases in {a.seconds:.0f} s (seed {a.seed}), {bad} mismatches; every case: repeat, first-k subset and "
      f"reversed batch give the same bits per pair")
sys.exit(1 if bad else 0)

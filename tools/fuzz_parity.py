#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (GPU box): random topology / batch / patch count, FR pairs and pairwise
triplets, against the oracle on the host: fp16x3 (the parity mode) at the north-star 1e-3, bf16x3 at 5e-3, fp16x2 (throughput mode) at a 2e-2 smoke bound.  Prints one line per case and a summary; exit code 1 on any miss."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vtamiq_amd import VTAMIQ, synth
from oracle import vtamiq_oracle as O
from tests.helpers import gate_error, split_inputs

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=24); ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--precision", default=None, help="force one numerics mode for every case (default: drawn per case)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.set_num_threads(16)
bad = 0
worst = 0.0
for ci in range(a.cases):
    variant = "ViT-L16" if rng.random() < 0.15 else "ViT-B16"
    L = int(rng.integers(1, 4))
    T = int(rng.choice([0, 0, 3, 8]))
    scales = int(rng.choice([0, 0, 2, 3]))
    B = int(rng.integers(1, 9)); N = int(rng.integers(4, 300))
    prec = str(rng.choice(["fp16x3", "fp16x3", "fp16x2", "bf16x3"]))
    if a.precision:
        prec = a.precision
    tol = 1e-3 if prec == "fp16x3" else (2e-2 if prec == "fp16x2" else 5e-3)     # fp16x2: throughput mode, smoke bound only (8e-3 seen)
    pairwise = rng.random() < 0.3
    kw = dict(vit_config=dict(variant=variant, num_keep_layers=L, num_extra_tokens=T, num_scales=scales, use_layer_scale=bool(T), pretrained=False),
              num_rgs=2, num_rcabs=2, calibrate=bool(rng.random() < 0.8), diff_scale=bool(rng.random() < 0.8))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=prec)
    sd = synth.make_state_dict(m.spec, 100 + ci)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
    patches, pos, sc = synth.make_inputs(m.spec, B, N, 200 + ci, aligned=bool(ci & 1))
    p, ps, s3 = split_inputs(patches, pos, sc, device="cuda")
    cp, cps, cs = split_inputs(patches, pos, sc)
    t = O.to_torch(sd)
    with torch.no_grad():
        if pairwise:
            # triplet (ref, dist, dist2) with dist2 = a second seeded distortion
            patches2, pos2, sc2 = synth.make_inputs(m.spec, B, N, 900 + ci, aligned=bool(ci & 1))
            p2, ps2, s32 = split_inputs(patches2, pos2, sc2, device="cuda")
            c2, cps2, cs2 = split_inputs(patches2, pos2, sc2)
            q = m.forward_pairwise((p[0], p[1], p2[1]), (ps[0], ps[1], ps2[1]), (s3[0], s3[1], s32[1]) if sc is not None else (None, None, None))
            q = torch.cat([t_.reshape(-1) for t_ in q[:2]]).cpu().numpy() if isinstance(q, (tuple, list)) else q.cpu().numpy().reshape(-1)
            qa = O.vtamiq_forward(t, m.spec, cp, cps, cs)[0].numpy()
            qb = O.vtamiq_forward(t, m.spec, (cp[0], c2[1]), (cps[0], cps2[1]), (cs[0], cs2[1]) if sc is not None else (None, None))[0].numpy()
            q_ref = np.concatenate([qa, qb])
        else:
            q = m(p, ps, s3)[0].cpu().numpy()
            q_ref = O.vtamiq_forward(t, m.spec, cp, cps, cs)[0].numpy()
    err = gate_error(q, q_ref)
    ok = np.isfinite(q).all() and err < tol
    bad += not ok
    worst = max(worst, float(err))
    print(f"case {ci:2d} {variant} L={L} T={T} scales={scales} B={B} N={N} {prec} pairwise={int(pairwise)} "
          f"calib={int(kw['calibrate'])}: err {err:.2e} {'ok' if ok else 'MISS'}", flush=True)
    del m; torch.cuda.empty_cache()
print("misses:", bad)
print("worst error:", f"{worst:.2e}")
sys.exit(1 if bad else 0)

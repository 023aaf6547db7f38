#!/usr/bin/env python3
"""Randomised stress of the two attention kernels (GPU box): for --seconds, random (sequences, S, heads, format, padding) cases on
random data with a few huge keys; every case runs the 4-wave kernel once and the software-pipelined kernel three times (a race in
its LDS ring or its asynchronous Q / output traffic would show as a run-to-run difference) and compares all outputs bit for bit,
plus the fp64 softmax on one (sequence, head).  Prints one line per 50 cases and a summary; exit code 1 on any mismatch."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=90.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
rng = random.Random(a.seed)
g = torch.Generator(device="cpu").manual_seed(a.seed)
t_end = time.time() + a.seconds
cases = bad = 0
worst = 0.0
while time.time() < t_end:
    H = rng.choice([768, 768, 1024])
    S = rng.choice([1, 2, 31, 33, 63, 64, 65, 127, 129, 255, 256, 257, 300, 501, 509, 521, 700, 1025, 1300]) if rng.random() < 0.8 else rng.randint(1, 1400)
    nseq = rng.choice([1, 2, 3, 5, 8, 13, 22, 40, 64]) if S <= 600 else rng.choice([1, 2, 4, 9, 16])
    fmt = rng.choice(["fp16x3", "fp16x3", "bf16x3", "fp16", "bf16"])
    S_pad = S if rng.random() < 0.6 else (S + 31) // 32 * 32
    rows = nseq * S_pad + 128
    qkv = torch.randn(rows, 3 * H, generator=g) * rng.choice([0.3, 1.0, 2.5])
    for _ in range(rng.randint(0, 3)):                                  # a few dominant keys
        qkv[rng.randrange(nseq) * S_pad + rng.randrange(S), H + 64 * rng.randrange(H // 64):][:64] *= rng.choice([4.0, 8.0])
    qkv = qkv.cuda()
    P = to_planes(qkv, fmt, "a")
    outs = []
    for variant in (0, 1, 2, 2):          # 4-wave kernel, pipelined kernel, split form twice (forced; S < 256 or a multiple of it: the pipelined kernel)
        lib.vtq_debug_attention_variant(variant)
        out = torch.full((P.shape[0], rows, H), 3.0, dtype=elt_dtype(fmt), device="cuda")
        _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S_pad, H, num_code(fmt), stream()))
        valid = out.view(P.shape[0], -1, H)[:, : nseq * S_pad].reshape(P.shape[0], nseq, S_pad, H)[:, :, :S]
        outs.append(valid.clone())
    lib.vtq_debug_attention_variant(-1)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:])
    sq, hd = rng.randrange(nseq), rng.randrange(H // 64)
    x = planes_value(P)[sq * S_pad: sq * S_pad + S].view(S, 3, H // 64, 64)[:, :, hd]
    ref = torch.softmax(x[:, 0] @ x[:, 1].t() / 8.0, -1) @ x[:, 2]
    got = planes_value(outs[1])[sq, :, hd * 64:(hd + 1) * 64]
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    tol = {"fp16x3": 2e-5, "bf16x3": 2e-4, "fp16": 4e-3, "bf16": 3e-2}[fmt]
    cases += 1
    worst = max(worst, err / tol)
    if not same or not err <= tol:
        bad += 1
        print(f"MISMATCH nseq={nseq} S={S} S_pad={S_pad} H={H} {fmt}: bit-identical {same}, err {err:.2e} (tol {tol:.0e})", flush=True)
        for i, j in ((0, 1), (1, 2), (2, 3)):
            d = (outs[i].view(torch.int16) != outs[j].view(torch.int16))
            if d.any():
                idx = d.nonzero()
                rows_ = sorted(set((int(r[1]), int(r[2])) for r in idx[:4000].tolist()))
                print(f"    run {i} vs run {j}: {int(d.sum())} elements differ; (sequence, query row) of the first: {rows_[:12]}; heads {sorted(set(int(r[3]) // 64 for r in idx[:4000].tolist()))[:12]}", flush=True)
    if cases % 50 == 0:
        print(f"{cases} cases, {bad} bad, worst err / tol {worst:.2f}", flush=True)
print(f"attention stress: {cases} random cases in {a.seconds:.0f} s (seed {a.seed}), {bad} mismatches; 4-wave = pipelined = split form (x2 runs) bit for bit in every case; "
      f"worst error against the fp64 softmax {worst:.2f} of the format's tolerance")
sys.exit(1 if bad else 0)

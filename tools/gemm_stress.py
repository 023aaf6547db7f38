#!/usr/bin/env python3
"""Randomised stress of the GEMM kernel (GPU box): for --seconds, random (M, N, K, operand format, epilogue) cases on random data.
Every case runs three times from the same inputs and the three outputs are compared bit for bit (a race in the DMA ring, in the
cross-tile chaining or in the staged epilogue would show as a run-to-run difference); the first M - 256 rows are also compared bit
for bit with a run on M - 256 rows only (the result of a row must not depend on the tile schedule, which changes with M), and a
row sample is checked against an fp64 product of the same rounded operands; and the case is run once more with every tile shape forced
(the persistent 256x256 kernel and the three small-tile shapes of csrc/gemm_st.hip; the first three runs use the library's rule) and compared
bit for bit -- the bitwise contract that makes the tile shape a pure speed choice.  Exit code 1 on any mismatch."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, planes_value, stream

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
rng = random.Random(a.seed)
g = torch.Generator(device="cpu").manual_seed(a.seed)
dev = "cuda"
t_end = time.time() + a.seconds
cases = bad = 0
worst = 0.0
while time.time() < t_end:
    fmt = rng.choice(["fp16x3", "fp16x3", "fp16x2", "fp16", "bf16x3", "bf16"])
    terms = FORMATS[fmt][1]
    kq = 128 if terms == 1 else 64
    M = 256 * rng.choice([2, 3, 5, 8, 13, 21, 34, 63, 126])
    N = 256 * rng.choice([1, 2, 3, 4, 6, 9, 12])
    K = kq * rng.randint(1, 3072 // kq) if rng.random() < 0.5 else rng.choice([768, 1024, 3072, 4096, 256])
    K = max(kq, K // kq * kq)
    epi = rng.choice([0, 1, 2])
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias, gamma = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    npl = Ap.shape[0]
    x0 = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None

    def run(m):
        out = torch.zeros((npl, M, N), dtype=elt_dtype(fmt), device=dev) if epi != 2 else None
        x = x0.clone() if epi == 2 else None
        _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, m, N, K, num_code(fmt), epi, bias.data_ptr(),
                                  gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                  out.data_ptr() if epi != 2 else None, M * N, N, stream()))
        return x if epi == 2 else out

    r1, r2, r3 = run(M), run(M), run(M)
    rs = run(M - 256)
    forced = []
    for v in (0, 1, 2, 3):
        if M * N > 256 * 256 * 400 and v in (1, 2):
            continue                              # 64x64 tiles on the largest cases: minutes of nothing new
        _lib.check(lib.vtq_debug_gemm_variant(v))
        forced.append((v, run(M)))
    _lib.check(lib.vtq_debug_gemm_variant(-1))
    torch.cuda.synchronize()
    bits = (lambda t: t.view(torch.int32)) if epi == 2 else (lambda t: t.view(torch.int16))
    same = torch.equal(bits(r1), bits(r2)) and torch.equal(bits(r1), bits(r3))
    tiles_ok = all(torch.equal(bits(r1), bits(rv)) for _, rv in forced)
    rowsel = (slice(None), slice(0, M - 256)) if epi != 2 else (slice(0, M - 256),)
    sched = torch.equal(bits(r1[rowsel]), bits(rs[rowsel]))
    rows = torch.tensor(sorted({0, 1, 255, 256, M // 2 + 3, M - 257, M - 129, M - 1}), device=dev)
    h = planes_value(Ap)[rows] @ planes_value(Wp).t() + bias.double()
    ref = torch.nn.functional.gelu(h) if epi == 1 else (x0[rows].double() + gamma.double() * h if epi == 2 else h)
    got = r1[rows].double() if epi == 2 else planes_value(r1)[rows]
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    tol = ({"bf16": 1.5e-2, "fp16": 2e-3, "fp16x2": 2e-3}.get(fmt, 1e-4)) if epi != 2 else ({"bf16": 1.5e-2, "fp16": 2e-3, "fp16x2": 2e-3}.get(fmt, 3e-5))
    cases += 1
    worst = max(worst, err / tol)
    if not (same and sched and tiles_ok and err <= tol):
        bad += 1
        print(f"MISMATCH {fmt} M={M} N={N} K={K} epilogue={epi}: three runs identical {same}, rows independent of M {sched}, every tile shape the same bits {tiles_ok} "
              f"{[v for v, rv in forced if not torch.equal(bits(r1), bits(rv))]}, err {err:.2e} (tol {tol:.0e})", flush=True)
    if cases % 25 == 0:
        print(f"{cases} cases, {bad} bad, worst err / tol {worst:.2f}", flush=True)
    del A, W, Ap, Wp, r1, r2, r3, rs, x0, forced
print(f"gemm stress: {cases} random cases in {a.seconds:.0f} s (seed {a.seed}), {bad} mismatches; every case: 3 runs bit-identical, rows bit-identical to a run "
      f"with one row panel less and to runs with every tile shape forced (256x256 persistent, 64x64 r3, 64x64 r2, 128x128), sampled rows against fp64 (worst {worst:.2f} of the format's tolerance)")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""In-kernel clock of the GEMM (MI355X_MICROARCH.md 'DVFS give-back' item 6) and the price of vector work in the MFMA shadow.

Needs the DIAGNOSTIC build of the library (tools/build_abl.sh diag "-DVTQ_GEMM_DIAG"; VTQ_LIB_PATH=tools/_abl/diag.so): it stamps
s_memtime / s_memrealtime around every K loop and around the whole kernel into a buffer nothing else reads.  The shipped library
executes no stamp.  Per --shadow value n the kernel also issues 8 n dummy v_fma_f32 in every LDS-read phase of the main loop.

    VTQ_LIB_PATH=tools/_abl/diag.so [VTQ_GEMM_CUS=8] [VTQ_GEMM_FLAGS=8] python3 tools/clock_probe.py --fmt fp16x3 fp16 --shadow 0 4 8 12
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32256)
ap.add_argument("--fmt", nargs="+", default=["fp16x3", "fp16"])
ap.add_argument("--only", nargs="+", default=["fc1"])
ap.add_argument("--shadow", type=int, nargs="+", default=[0])
ap.add_argument("--warm", type=float, default=2.0, help="seconds of back-to-back launches before the stamped ones")
ap.add_argument("--timed", type=int, default=40)
a = ap.parse_args()
lib = _lib.load()
dev = "cuda"
M = a.M
shapes = {"qkv": (2304, 768, 0), "outproj": (768, 768, 2), "fc1": (3072, 768, 1), "fc2": (768, 3072, 2)}
g = torch.Generator(device="cpu").manual_seed(0)
diag = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
is_diag = lib.vtq_debug_gemm_diag(diag.data_ptr(), 0)
tag = f"flags={os.environ.get('VTQ_GEMM_FLAGS', '0')} cus={os.environ.get('VTQ_GEMM_CUS', '32')} sched={os.environ.get('VTQ_GEMM_SCHED', '-')}"
if not is_diag:
    print("# NOT a diagnostic build: no stamps; timing only", flush=True)

for fmt in a.fmt:
    for name in a.only:
        N, K, epi = shapes[name]
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
        bias, gamma = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
        Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
        out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=dev) if epi != 2 else None
        x = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None

        def call():
            _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), epi, bias.data_ptr(),
                                      gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                      out.data_ptr() if epi != 2 else None, M * N, N, stream()))
        for sh in a.shadow:
            lib.vtq_debug_gemm_diag(diag.data_ptr(), sh)
            call()
            torch.cuda.synchronize()
            t0 = time.time()
            while time.time() - t0 < a.warm:             # >= 2 s of back-to-back launches on random data
                for _ in range(50):
                    call()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.timed):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.timed * 1e3
            line = f"{tag} {fmt:7s} {name:8s} M={M} shadow={8 * sh:4d} VALU/phase: {us:8.1f} us/launch"
            if is_diag:
                full = diag.view(256, 64).cpu()
                full = full[full[:, 4] > 0].double()
                d = full[:, :8]
                ew = full[:, 8:40].view(-1, 8, 4)                # per wave: convert, copy-out, interval-end wait (cycles, summed over tiles)
                loop_ghz = (d[:, 0] / d[:, 1] * 0.1)          # memrealtime ticks at 100 MHz
                kern_ghz = (d[:, 2] / d[:, 3] * 0.1)
                kern_us = d[:, 3] / 100.0
                loop_share = d[:, 0] / d[:, 2]
                line += (f"  in-kernel clock: K loops {loop_ghz.median():.3f} GHz (min {loop_ghz.min():.3f} max {loop_ghz.max():.3f}), "
                         f"whole kernel {kern_ghz.median():.3f} GHz; workgroup lifetime median {kern_us.median():.1f} us max {kern_us.max():.1f} us; "
                         f"K-loop share of cycles {loop_share.median():.3f}; workgroups {d.shape[0]}")
                tiles = d[:, 4].median().item()
                for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
                    c = ew[:, sl, :].mean(dim=1).median(dim=0).values / max(tiles, 1)
                    line += f"\n    epilogue per tile, {grp}: convert {c[0]:.0f} cyc, copy-out {c[1]:.0f} cyc, interval waits {c[2]:.0f} cyc (stamps included)"
            print(line, flush=True)
        del A, W, Ap, Wp, out, x

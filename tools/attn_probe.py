#!/usr/bin/env python3
"""Where a wave's cycles go in the two fused attention kernels, and at what clock (GPU box).

With the shipped library: sustained time per launch (>= --warm seconds of back-to-back launches, then --timed launches).
With the diagnostic build (tools/build_abl.sh adiag "-DVTQ_ATTN_DIAG"; VTQ_LIB_PATH=tools/_abl/adiag.so) every wave accumulates s_memtime
spans of the phases of a key tile and of its lifetime, plus s_memrealtime for the clock, into a buffer nothing else reads:
4-wave kernel: stage issue | QK^T | softmax | PV | end-of-tile wait + barrier; pipelined kernel: phase 1 | phase 2 | seam + copies |
wait + barrier.  Every stamp costs an `s_waitcnt lgkmcnt(0)` and a scalar-memory round trip (about 300 cycles): spans are upper bounds.
The skeleton switches of the pipelined kernel (-DVTQ_SW_NOFILL / NOMFMA / NOSTORE / NOQ / NODMA = 1, results wrong by design) are timed the
same way; profiles/r03_attention_anatomy.txt holds the numbers.

    VTQ_LIB_PATH=tools/_abl/adiag.so [VTQ_ATTN_LDS_PAD=bytes] python3 tools/attn_probe.py --variant 0|1 --fmt fp16x3 fp16
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, stream

ap = argparse.ArgumentParser()
ap.add_argument("--nseq", type=int, default=64)
ap.add_argument("--S", type=int, default=501)
ap.add_argument("--H", type=int, default=768)
ap.add_argument("--fmt", nargs="+", default=["fp16x3", "fp16"])
ap.add_argument("--warm", type=float, default=1.0)
ap.add_argument("--timed", type=int, default=50)
ap.add_argument("--tag", default=os.environ.get("VTQ_LIB_PATH", "shipped"))
ap.add_argument("--variant", type=int, default=-1, help="0 = 4-wave kernel, 1 = pipelined kernel, -1 = the library's rule")
a = ap.parse_args()
lib = _lib.load()
if a.variant >= 0:
    lib.vtq_debug_attention_variant(a.variant)
rows = a.nseq * a.S + 128
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(rows, 3 * a.H, generator=g) * 1.5).cuda()
nwaves = max(((a.S + 127) // 128) * (a.H // 64) * a.nseq * 4, 256 * 8)
diag = torch.zeros(max(256 * 64, nwaves * 16), dtype=torch.int64, device="cuda")
if hasattr(lib, "vtq_debug_gemm_diag"):
    lib.vtq_debug_gemm_diag(diag.data_ptr(), 0)
for fmt in a.fmt:
    P = to_planes(qkv, fmt, "a")
    out = torch.zeros((P.shape[0], rows, a.H), dtype=elt_dtype(fmt), device="cuda")
    call = lambda: _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * a.H, out.data_ptr(), rows * a.H, a.nseq, a.S, a.S, a.H, num_code(fmt), stream()))
    call(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < a.warm:
        for _ in range(50):
            call()
        torch.cuda.synchronize()
    diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.timed):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.timed * 1e3
    dw = diag[:nwaves * 16].view(nwaves, 16).cpu().double()
    dw = dw[dw[:, 6] > 0]
    d = dw.sum(dim=0)
    line = f"{a.tag} pad={os.environ.get('VTQ_ATTN_LDS_PAD', '0')} attention {fmt} nseq={a.nseq} S={a.S}: {us:7.1f} us/launch"
    if d[6] > 0 and a.variant == 1:
        waves, tiles = d[6].item(), d[7].item()
        ghz = d[0] / d[1] * 0.1
        per = lambda k: d[k].item() / tiles
        line += (f"  [pipelined] in-kernel clock {ghz:.3f} GHz; {waves:.0f} waves, lifetime {d[0].item() / waves:.0f} cyc = {d[1].item() / waves / 100:.1f} us, "
                 f"{tiles / waves:.0f} tiles each; per wave and tile: phase 1 (QK^T || split) {per(2):.0f} | phase 2 (PV || softmax) {per(3):.0f} | seam / copies {per(4):.0f} | "
                 f"wait + barrier {per(5):.0f} cyc (lifetime/tiles {d[0].item() / tiles:.0f}); prologue {d[9].item() / waves:.0f} cyc per wave")
        nk = dw[:, 13].long()
        n0, n1, n2 = (nk & 0xFFFFF).sum().item(), ((nk >> 20) & 0xFFFFF).sum().item(), ((nk >> 40) & 0xFFFFF).sum().item()
        if n0 + n1 + n2 > 0:
            line += (f"\n    whole iterations by kind (cycles per wave): plain {d[10].item() / max(n0, 1):.0f} (x{n0 / waves:.1f} per wave) | writing the previous block's output "
                     f"{d[11].item() / max(n1, 1):.0f} (x{n1 / waves:.1f}; write_block itself {d[14].item() / max(n1, 1):.0f}) | loading the next block's Q {d[12].item() / max(n2, 1):.0f} (x{n2 / waves:.1f}); "
                     f"after the last iteration (final output) {d[15].item() / waves:.0f}")
        for grp in (0, 1):
            sel = dw.view(-1, 8, 16)[:, 4 * grp:4 * grp + 4].reshape(-1, 16).sum(dim=0)
            t = sel[7].item()
            line += f"\n    waves {4 * grp}-{4 * grp + 3}: phase 1 {sel[2].item() / t:.0f} | phase 2 {sel[3].item() / t:.0f} | seam / copies {sel[4].item() / t:.0f} | wait + barrier {sel[5].item() / t:.0f}"
    elif d[6] > 0:
        waves, tiles = d[6].item(), d[7].item()
        ghz = d[0] / d[1] * 0.1
        per = lambda k: d[k].item() / tiles
        line += (f"  in-kernel clock {ghz:.3f} GHz; wave lifetime {d[0].item() / waves:.0f} cyc = {d[1].item() / waves / 100:.1f} us; per wave and key tile: "
                 f"stage issue {per(8):.0f} | QK^T {per(2):.0f} | softmax {per(3):.0f} | PV {per(4):.0f} | wait+barrier {per(5):.0f} cyc"
                 f" (sum {sum(per(k) for k in (8, 2, 3, 4, 5)):.0f}; lifetime/tiles {d[0].item() / tiles:.0f})")
        line += f"; per wave: prologue (start .. first tile) {d[9].item() / waves:.0f} cyc, epilogue (last tile .. stores issued) {d[10].item() / waves:.0f} cyc"
    print(line, flush=True)

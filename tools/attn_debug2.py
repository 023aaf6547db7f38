#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtamiq_amd import _lib
from tests.gpu_util import elt_dtype, num_code, to_planes, planes_value, stream
lib = _lib.load()
nseq, S, H = 4, 501, 768
rows = nseq * S + 128
g = torch.Generator(device="cpu").manual_seed(16)
x = (torch.randn(rows, 3 * H, generator=g) * 1.5).half().float()
fmt = "fp16x3"
P = to_planes(x.cuda(), fmt, "a")
out = torch.zeros((2, rows, H), dtype=torch.float16, device="cuda")
_lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
torch.cuda.synchronize()
nh = H // 64
xx = planes_value(P)[: nseq * S].view(nseq, S, 3, nh, 64)
q, k, v = (xx[:, :, i].permute(0, 2, 1, 3) for i in range(3))
ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(nseq * S, H)
hi, lo = out[0][: nseq * S].double(), out[1][: nseq * S].double()
d = (hi + lo - ref).abs()
print("max|ref|", ref.abs().max().item(), "max err", d.max().item())
idx = torch.nonzero(d > 0.2 * d.max())
print("n outliers", len(idx), "of", d.numel())
for r, c in idx[:25].tolist():
    print(f"row {r} (seq {r // S}, tok {r % S}) col {c} (head {c // 64}, d {c % 64}): ref {ref[r, c].item():+.7f} hi {hi[r, c].item():+.7f} lo {lo[r, c].item():+.3e} err {d[r, c].item():.2e}  ref-hi {ref[r, c].item() - hi[r, c].item():+.3e}")
toks = (idx[:, 0] % S)
print("token positions of outliers: min", toks.min().item(), "max", toks.max().item(), "unique", toks.unique()[:20].tolist())
print("d-cols:", (idx[:, 1] % 64).unique().tolist()[:40])

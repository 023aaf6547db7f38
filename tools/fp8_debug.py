#!/usr/bin/env python3
"""Stage-by-stage comparison of the engine's fp8 mode with the fake-quant oracle on layer `--layer` of a golden case (GPU box).
Uses the test hooks vtq_debug_stop_after / vtq_debug_buffers."""
import argparse, ctypes as C, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import fp8_oracle as F8, vtamiq_oracle as O
from tests.helpers import load_case, split_inputs
from vtamiq_amd import VTAMIQ, _lib
from vtamiq_amd.experimental_fp8 import model_class      # VTAMIQFp8 for "fp8" (a build of the experiment), VTAMIQ otherwise

ap = argparse.ArgumentParser(); ap.add_argument("--case", default="c1_b2_n50"); ap.add_argument("--layer", type=int, default=0)
a = ap.parse_args()
g, kw, spec, sd, (patches, pos, scales) = load_case(a.case)
m = model_class("fp8")(**json.loads(json.dumps(kw)), precision="fp8", engine_options=_lib.OPT_FP8_STATIC_SCALES)   # the oracle's static scales (F8.S_*)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m = m.cuda().eval()
p, ps, sc = split_inputs(patches, pos, scales, device="cuda")
lib = _lib.load(); hip = C.CDLL("libamdhip64.so")
with torch.no_grad(): m(p, ps, sc)          # creates the engine
B, N = int(g["B"]), int(g["N"]); H, Md = spec.hidden_size, spec.mlp_dim
S = N + spec.num_tokens; S_pad = S; nseq = 2 * B      # sequences are packed back to back (engine.hip geometry())


def grab(stage):
    _lib.check(lib.vtq_debug_stop_after(m._engine, a.layer * 7 + stage))
    with torch.no_grad(): m(p, ps, sc)
    torch.cuda.synchronize()
    x, ln, big, rows = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    _lib.check(lib.vtq_debug_buffers(m._engine, C.byref(x), C.byref(ln), C.byref(big), C.byref(rows)))
    R = rows.value
    def copy(ptr, nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), ptr, C.c_size_t(nbytes), 3) == 0
        return t.cpu()
    return copy(x, R * H * 4).view(torch.float32).view(R, H), copy(ln, R * H * 2), copy(big, R * max(3 * H, Md) * 2), R


def seqs(t):      # [rows, W] -> [nseq, S, W]
    return t[:nseq * S_pad].view(nseq, S_pad, -1)[:, :S]


def e4(b): return b.view(torch.float8_e4m3fn).float()
def rep(name, got, want):
    d = (got - want).abs()
    print(f"{name:10s} max|d| {d.max():.3e} / max|want| {want.abs().max():.3e}   mismatching elements {(d > 1e-6 * want.abs().max()).float().mean():.3e}", flush=True)


# oracle stages (fp32), layer a.layer input = the oracle's own stream
sdt = O.to_torch(sd)
pc, psc, scc = split_inputs(patches, pos, scales)
xs = []
for i in range(2):
    x = F8.embeddings(sdt, spec, pc[i], psc[i], scc[i])
    for l in range(a.layer): x = F8.encoder_layer(sdt, spec, l, x)
    xs.append(x)
x = torch.cat(xs)                       # [2B, S, H]
pre = f"transformer.encoder.layers.{a.layer}."
x_gpu, ln, _, R = grab(0)              # LayerNorm 1 leaves the stream untouched: x is the layer's input
rep("x in", seqs(x_gpu), x)
ln8 = F8.quant_act(O._layer_norm(x, sdt[pre + "attention_norm.weight"], sdt[pre + "attention_norm.bias"]), F8.S_LN)
rep("ln1 (x8)", seqs(e4(ln[:R * H]).view(R, H)), ln8)
qkv = torch.cat([F8.linear8(ln8, F8.S_LN, sdt[f"{pre}attn.{n}.weight"], sdt[f"{pre}attn.{n}.bias"]) for n in ("query", "key", "value")], -1)
_, _, big, _ = grab(1); rep("qkv", seqs(big[:R * 3 * H * 2].view(torch.float16).float().view(R, 3 * H)), qkv)
nh, dh = spec.num_heads, H // spec.num_heads
q, k, v = (t.view(nseq, S, nh, dh).permute(0, 2, 1, 3) for t in qkv.split(H, -1))
ctx = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).permute(0, 2, 1, 3).reshape(nseq, S, H)
ctx8 = F8.quant_act(ctx, F8.S_ATT)
_, ln, _, _ = grab(2); rep("ctx (x16)", seqs(e4(ln[:R * H]).view(R, H)), ctx8)
h = F8.linear8(ctx8, F8.S_ATT, sdt[pre + "attn.out.weight"], sdt[pre + "attn.out.bias"])
if spec.use_layer_scale: h = h * sdt[pre + "ls1.gamma"]
x1 = x + h
xg, _, _, _ = grab(3); rep("x + attn", seqs(xg), x1)
ln8 = F8.quant_act(O._layer_norm(x1, sdt[pre + "ffn_norm.weight"], sdt[pre + "ffn_norm.bias"]), F8.S_LN)
_, ln, _, _ = grab(4); rep("ln2 (x8)", seqs(e4(ln[:R * H]).view(R, H)), ln8)
g8 = F8.quant_act(F.gelu(F8.linear8(ln8, F8.S_LN, sdt[pre + "ffn.fc1.weight"], sdt[pre + "ffn.fc1.bias"])), F8.S_GELU)
_, _, big, _ = grab(5); rep("gelu (x4)", seqs(e4(big[:R * Md]).view(R, Md)), g8)
h = F8.linear8(g8, F8.S_GELU, sdt[pre + "ffn.fc2.weight"], sdt[pre + "ffn.fc2.bias"])
if spec.use_layer_scale: h = h * sdt[pre + "ls2.gamma"]
xg, _, _, _ = grab(6); rep("x + mlp", seqs(xg), x1 + h)

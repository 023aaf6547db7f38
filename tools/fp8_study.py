#!/usr/bin/env python3
"""Can an e4m3 mode be a SCORING mode?  (one bounded attempt.)  CPU study on the oracle, in the reference's own metric:
e4m3 operands only in fc1 / fc2 (60 % of the flops), everything else as in the fp16 single-plane mode ("h1") or the parity mode ("h3"), with
the activation scale taken per tensor, per row (computed by the producing epilogue), or per 32-element block (the E8M0 block scales of the
MX-scaled MFMA); weights per output channel or per block.  64 synthetic pairs on the bench's distortion ladder (N patches), flat-init and
trained-like (stress_state qk = 5) weights; target = the fp32 oracle's scores; SROCC / KROCC / PLCC via scipy.

    python tools/fp8_study.py [--patches 500] [--pairs 64]        (CPU; ~1 min per row at N = 500 on 8 cores)
"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F, scipy.stats
from oracle import vtamiq_oracle as O
from tests import numerics_study as NS
from tests.helpers import stress_state
from vtamiq_amd import synth
from vtamiq_amd.spec import make_spec

ap = argparse.ArgumentParser()
ap.add_argument("--patches", type=int, default=500); ap.add_argument("--images", type=int, default=8); ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--weights", nargs="+", default=["flat", "stress5"])
a = ap.parse_args()
torch.set_num_threads(a.threads)
E4 = torch.float8_e4m3fn


def q8(v, kdim, mode):
    """e4m3 rounding of v with power-of-two scales: 'tensor' one scale, 'row' one per slice along every axis but kdim's (i.e. per row of the
    contraction), 'block' one per 32 elements along kdim (MX)."""
    if mode == "block":
        return NS._mx8(v, kdim)
    if mode == "tensor":
        amax = v.abs().max().clamp_min(1e-30)
    else:
        amax = v.abs().amax(dim=kdim, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
    return (v * sc).clamp(-448, 448).to(E4).float() / sc


def emm8(a_, w_t, amode, wmode):
    return q8(a_, -1, amode) @ q8(w_t, -2, wmode)


def vit_tokens(sd, spec, patches, pos, base, amode, wmode, sites):
    """numerics_study.vit_tokens with the fc1 / fc2 (and optionally qkv / out) contractions on e4m3 operands"""
    sch = NS.scheme(base)
    B, N = patches.shape[:2]
    e = "transformer.embeddings."
    W = sd[e + "patch_embeddings.weight"]; H = W.shape[0]
    x = (NS.emm(patches.reshape(B * N, -1), W.reshape(H, -1).t(), sch["patch"]) + sd[e + "patch_embeddings.bias"]).view(B, N, H)
    table = sd[e + "positional_embeddings.positional_embeddings"][0]
    x = x + table[O.pos_index(pos.reshape(B * N, 2), spec.pos_grid)].view(B, N, -1)
    x = torch.cat([sd[e + "cls_token"].expand(B, 1, -1) + table[0], x], dim=1)
    nh = spec.num_heads; dh = H // nh
    mm = lambda site, a_, w_t: emm8(a_, w_t, amode, wmode) if site in sites else NS.emm(a_, w_t, sch[site])
    for i in range(spec.num_layers):
        p = f"transformer.encoder.layers.{i}."
        h = O._layer_norm(x, sd[p + "attention_norm.weight"], sd[p + "attention_norm.bias"]); S = h.shape[1]
        proj = lambda nm: (mm("qkv", h, sd[f"{p}attn.{nm}.weight"].t()) + sd[f"{p}attn.{nm}.bias"]).view(B, S, nh, dh).permute(0, 2, 1, 3)
        q, k, v = proj("query"), proj("key"), proj("value")
        scores = NS.emm(q, k.transpose(-1, -2), sch["qk"]) / math.sqrt(dh)
        pexp = torch.exp(scores - scores.max(dim=-1, keepdim=True).values)
        ctx = (NS.emm(pexp, v, sch["pv"]) / pexp.sum(dim=-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, S, H)
        x = x + mm("out", ctx, sd[p + "attn.out.weight"].t()) + sd[p + "attn.out.bias"]
        h = O._layer_norm(x, sd[p + "ffn_norm.weight"], sd[p + "ffn_norm.bias"])
        g = F.gelu(mm("fc1", h, sd[p + "ffn.fc1.weight"].t()) + sd[p + "ffn.fc1.bias"])
        x = x + mm("fc2", g, sd[p + "ffn.fc2.weight"].t()) + sd[p + "ffn.fc2.bias"]
    x = O._layer_norm(x, sd["transformer.encoder.encoder_norm.weight"], sd["transformer.encoder.encoder_norm.bias"])
    return x[:, :1]


ROWS = [("fp16 everywhere (the `fp16` mode)", "h1", None, None, ()),
        ("fc1+fc2 e4m3: act per TENSOR, w per channel; rest fp16", "h1", "tensor", "row", ("fc1", "fc2")),
        ("fc1+fc2 e4m3: act per ROW, w per channel; rest fp16", "h1", "row", "row", ("fc1", "fc2")),
        ("fc1+fc2 e4m3: act per 32-BLOCK (MX), w per block; rest fp16", "h1", "block", "block", ("fc1", "fc2")),
        ("fc1+fc2 e4m3: act per ROW, w per channel; rest fp16x3", "h3", "row", "row", ("fc1", "fc2")),
        ("fc1+fc2 e4m3: MX blocks; rest fp16x3", "h3", "block", "block", ("fc1", "fc2")),
        ("fc1 only e4m3 MX blocks; rest fp16x3", "h3", "block", "block", ("fc1",)),
        ("all four linears e4m3 MX blocks; attention fp16 (today's fp8 mode with block scales)", "h1", "block", "block", ("qkv", "out", "fc1", "fc2")),
        ("all four linears e4m3 act per tensor, w per channel; attention fp16 (today's fp8 mode)", "h1", "tensor", "row", ("qkv", "out", "fc1", "fc2"))]

spec = make_spec(dict(variant="ViT-B16"))
patches, pos, _ = synth.make_ladder_inputs(spec, a.images, a.patches, 777)
tp, tq = torch.from_numpy(patches), torch.from_numpy(pos)
print(f"# {patches.shape[0]} pairs ({a.images} images x 8 distortion strengths), N = {a.patches}, ViT-B/16 L = 12; target = fp32 oracle; columns: SROCC KROCC PLCC | worst raw rel. error over |q| >= 0.1 rms")
for wname in a.weights:
    sd_np = synth.make_state_dict(spec, 0) if wname == "flat" else stress_state(spec, 0, qk=float(wname.replace("stress", "")))
    sd = O.to_torch(sd_np)
    with torch.no_grad():
        q_ref = torch.cat([O.vtamiq_forward(sd, spec, (tp[i:i + 8, 0], tp[i:i + 8, 1]), (tq[i:i + 8, 0], tq[i:i + 8, 1]), (None, None))[0] for i in range(0, tp.shape[0], 8)]).numpy()
    rms = float(np.sqrt(np.mean(q_ref ** 2)))
    print(f"weights = {wname}: rms(q_ref) = {rms:.4e}")
    for name, base, am, wm, sites in ROWS:
        t0 = time.time()
        qs = []
        with torch.no_grad():
            for i in range(0, tp.shape[0], 8):
                tr = vit_tokens(sd, spec, tp[i:i + 8, 0], tq[i:i + 8, 0], base, am, wm, sites)
                td = vit_tokens(sd, spec, tp[i:i + 8, 1], tq[i:i + 8, 1], base, am, wm, sites)
                qs.append(O.head(sd, spec, tr, td))
        q = torch.cat(qs).numpy()
        big = np.abs(q_ref) >= 0.1 * rms
        err = float(np.max(np.abs(q - q_ref)[big] / np.abs(q_ref[big])))
        print(f"  {name:88s} {scipy.stats.spearmanr(q_ref, q)[0]:.6f} {scipy.stats.kendalltau(q_ref, q)[0]:.5f} {scipy.stats.pearsonr(q_ref, q)[0]:.6f} | {err:.2e}   ({time.time() - t0:.0f} s)", flush=True)

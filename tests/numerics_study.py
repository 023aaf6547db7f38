#!/usr/bin/env python3
"""Operand-format study on the CPU oracle (test infrastructure; not collected by pytest).

Which GEMM operand formats keep the pair score within the north-star 1e-3 (raw relative) of the fp32 reference?
Every dense contraction of the path is emulated with operands rounded the way an MFMA would see them and fp32
accumulation; everything else (LayerNorm, softmax, GELU, residual stream, DiffNet head) stays fp32 as in the engine.

A scheme assigns every contraction site a format string:
    "x"      exact fp32 operands (the oracle)
    "b1"     single bf16:  a_hi*w_hi                                   1 MFMA / product
    "b3"     bf16 hi/lo split: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo       3
    "h1"     single fp16                                               1
    "h2a"    fp16, activation split, weight single:  (a_hi + a_lo)*w_hi   2
    "h2w"    fp16, weight split, activation single:  a_hi*(w_hi + w_lo)   2
    "h3"     fp16 hi/lo split, three terms                             3
    "b2a"/"b2w"  the bf16 two-term forms
Sites: patch, qkv (q, k, v projections), qk (QK^T: a = q, w = k), pv (a = P, w = v), out, fc1, fc2.
`hi = fmt(v)`, `lo = fmt(v - hi)`; fp16 subnormals are kept (torch CPU semantics).

    python tests/numerics_study.py [--cases ...] [--stress]        -> table on stdout (DESIGN.md section 2 quotes it)
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vtamiq_oracle as O          # noqa: E402
from tests import helpers                     # noqa: E402

SITES = ["patch", "qkv", "qk", "pv", "out", "fc1", "fc2"]
MFMAS = {"m2": 2, "m2t": 2, "m15": 1.5, "m15t": 1.5, "m25": 2.5, "m25t": 2.5, "x": 0, "b1": 1, "h1": 1, "b2a": 2, "b2w": 2, "h2a": 2, "h2w": 2, "b3": 3, "h3": 3}


def _split(v, dt):
    hi = v.to(dt).float()
    lo = (v - hi).to(dt).float()
    return hi, lo


def _mx8(v, kdim, block=32, tensor_scale=False):
    """e4m3 rounding of v with one power-of-two scale per `block` elements along the contraction axis `kdim` (what the MX-scaled
    MFMA's E8M0 operands provide), or one per tensor: the largest 2^k with max|v| * 2^k <= 448."""
    v = v.movedim(kdim, -1)
    K = v.shape[-1]
    if tensor_scale:
        amax = v.abs().max().clamp_min(1e-30).expand(1)
        sc = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
        r = (v * sc).to(torch.float8_e4m3fn).float() / sc
    else:
        pad = (-K) % block
        vp = F.pad(v, (0, pad)).reshape(*v.shape[:-1], -1, block)
        amax = vp.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        sc = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
        r = ((vp * sc).to(torch.float8_e4m3fn).float() / sc).reshape(*v.shape[:-1], -1)[..., :K]
    return r.movedim(-1, kdim)


def emm(a, w_t, fmt):
    """a @ w_t with operands rounded per `fmt` (w_t already oriented [K, N] or batched)."""
    if fmt == "x":
        return a @ w_t
    if fmt[0] == "m":
        # fp16 hi*hi on the f16 MFMA + cross terms on the double-rate MX-scaled e4m3 MFMA:
        #   m2  : + e4m3(a_hi) e4m3(w_lo) + e4m3(a_lo) e4m3(w_hi)        1 + 0.5 + 0.5 = 2 MFMA-equivalents / product
        #   m15 : + e4m3(a_lo) e4m3(w_hi)  (weight lo dropped)            1.5
        #   m25 : + a_lo w_hi in fp16 + e4m3(a_hi) e4m3(w_lo)             2.5
        # suffix t: per-tensor instead of per-32-block scales
        ts = fmt.endswith("t")
        kind = fmt[1:].rstrip("t")
        ah, al = _split(a, torch.float16)
        wh, wl = _split(w_t, torch.float16)
        q = lambda v, kd: _mx8(v, kd, tensor_scale=ts)
        y = ah @ wh
        if kind == "2":
            y = y + q(a, -1) @ q(wl, -2) + q(al, -1) @ q(w_t, -2)
        elif kind == "15":
            y = y + q(al, -1) @ q(w_t, -2)
        elif kind == "25":
            y = y + al @ wh + q(a, -1) @ q(wl, -2)
        else:
            raise ValueError(fmt)
        return y
    dt = torch.bfloat16 if fmt[0] == "b" else torch.float16
    kind = fmt[1:]
    ah, al = _split(a, dt)
    wh, wl = _split(w_t, dt)
    y = ah @ wh
    if kind in ("2a", "3"):
        y = y + al @ wh
    if kind in ("2w", "3"):
        y = y + ah @ wl
    return y


def flops_weights(spec, N):
    """share of the pair's dense flops per site (SURVEY 8d flop model)."""
    H, M, L, S = spec.hidden_size, spec.mlp_dim, spec.num_layers, spec.seq_len(N)
    f = {"patch": 2.0 * N * spec.patch_dim * H, "qkv": L * 6.0 * S * H * H, "qk": L * 2.0 * S * S * H, "pv": L * 2.0 * S * S * H,
         "out": L * 2.0 * S * H * H, "fc1": L * 2.0 * S * H * M, "fc2": L * 2.0 * S * H * M}
    tot = sum(f.values())
    return {k: v / tot for k, v in f.items()}


def vit_tokens(sd, spec, patches, pos, scales, sch):
    B, N = patches.shape[:2]
    e = "transformer.embeddings."
    W = sd[e + "patch_embeddings.weight"]
    H = W.shape[0]
    x = (emm(patches.reshape(B * N, -1), W.reshape(H, -1).t(), sch["patch"]) + sd[e + "patch_embeddings.bias"]).view(B, N, H)
    table = sd[e + "positional_embeddings.positional_embeddings"][0]
    x = x + table[O.pos_index(pos.reshape(B * N, 2), spec.pos_grid)].view(B, N, -1)
    if spec.use_scale_embedding:
        st = sd[e + "scale_embeddings.scale_embeddings"][0]
        x = x + st[O.scale_index(scales.reshape(B * N), spec.num_scales)].view(B, N, -1)
    toks = [sd[e + "cls_token"].expand(B, 1, -1) + table[0]]
    if spec.num_extra_tokens > 0:
        toks.append(sd[e + "extra_tokens"].expand(B, spec.num_extra_tokens, -1))
    x = torch.cat(toks + [x], dim=1)
    nh = spec.num_heads
    dh = H // nh
    for i in range(spec.num_layers):
        p = f"transformer.encoder.layers.{i}."
        fm = lambda site: sch.get(f"{site}@{i}", sch[site])          # per-layer override "site@layer"
        h = O._layer_norm(x, sd[p + "attention_norm.weight"], sd[p + "attention_norm.bias"])
        S = h.shape[1]

        def proj(nm):
            y = emm(h, sd[f"{p}attn.{nm}.weight"].t(), fm("qkv")) + sd[f"{p}attn.{nm}.bias"]
            return y.view(B, S, nh, dh).permute(0, 2, 1, 3)
        q, k, v = proj("query"), proj("key"), proj("value")
        scores = emm(q, k.transpose(-1, -2), fm("qk")) / math.sqrt(dh)
        # the engine's P is exp(s - max) un-normalised (values in (0, 1]); the row sum divides O afterwards
        mx = scores.max(dim=-1, keepdim=True).values
        pexp = torch.exp(scores - mx)
        ctx = emm(pexp, v, fm("pv")) / pexp.sum(dim=-1, keepdim=True)
        ctx = ctx.permute(0, 2, 1, 3).reshape(B, S, H)
        a = emm(ctx, sd[p + "attn.out.weight"].t(), fm("out")) + sd[p + "attn.out.bias"]
        if spec.use_layer_scale:
            a = a * sd[p + "ls1.gamma"]
        x = x + a
        h = O._layer_norm(x, sd[p + "ffn_norm.weight"], sd[p + "ffn_norm.bias"])
        g = F.gelu(emm(h, sd[p + "ffn.fc1.weight"].t(), fm("fc1")) + sd[p + "ffn.fc1.bias"])
        m = emm(g, sd[p + "ffn.fc2.weight"].t(), fm("fc2")) + sd[p + "ffn.fc2.bias"]
        if spec.use_layer_scale:
            m = m * sd[p + "ls2.gamma"]
        x = x + m
    x = O._layer_norm(x, sd["transformer.encoder.encoder_norm.weight"], sd["transformer.encoder.encoder_norm.bias"])
    return x[:, :spec.num_tokens]


@torch.no_grad()
def forward(sd, spec, inputs, sch):
    p, ps, sc = inputs
    t_ref = vit_tokens(sd, spec, p[0], ps[0], sc[0], sch)
    t_dist = vit_tokens(sd, spec, p[1], ps[1], sc[1], sch)
    return O.head(sd, spec, t_ref, t_dist).numpy()


def scheme(default, **over):
    s = {k: default for k in SITES}
    s.update({k.replace("_at_", "@"): v for k, v in over.items()})
    return s


SCHEMES = [
    ("exact", scheme("x")),
    ("bf16 (1 MFMA)", scheme("b1")),
    ("bf16x3 everywhere (round-1 parity mode)", scheme("b3")),
    ("fp16 (1 MFMA)", scheme("h1")),
    ("fp16x3 everywhere", scheme("h3")),
    ("fp16 a-split, w single (2)", scheme("h2a")),
    ("fp16 w-split, a single (2)", scheme("h2w")),
    ("bf16 a-split, w single (2)", scheme("b2a")),
    ("bf16 w-split, a single (2)", scheme("b2w")),
    # which site tolerates what: one site downgraded from the 3-term bf16 form
    ("b3, patch b1", scheme("b3", patch="b1")),
    ("b3, pv b2w (P single, V split)", scheme("b3", pv="b2w")),
    ("b3, pv b2a (P split, V single)", scheme("b3", pv="b2a")),
    ("b3, qk b1", scheme("b3", qk="b1")),
    ("b3, fc1 b1", scheme("b3", fc1="b1")),
    ("b3, fc2 b1", scheme("b3", fc2="b1")),
    ("b3, qkv b1", scheme("b3", qkv="b1")),
    ("b3, out b1", scheme("b3", out="b1")),
    # ... and from the 3-term fp16 form
    ("h3, patch h2a", scheme("h3", patch="h2a")),
    ("h3, patch h1", scheme("h3", patch="h1")),
    ("h3, qkv h2a", scheme("h3", qkv="h2a")),
    ("h3, qkv h2w", scheme("h3", qkv="h2w")),
    ("h3, qk h2a (Q split, K single)", scheme("h3", qk="h2a")),
    ("h3, qk h1", scheme("h3", qk="h1")),
    ("h3, pv h2w (P single, V split)", scheme("h3", pv="h2w")),
    ("h3, pv h2a (P split, V single)", scheme("h3", pv="h2a")),
    ("h3, pv h1", scheme("h3", pv="h1")),
    ("h3, out h2a", scheme("h3", out="h2a")),
    ("h3, out h2w", scheme("h3", out="h2w")),
    ("h3, fc1 h2a", scheme("h3", fc1="h2a")),
    ("h3, fc1 h2w", scheme("h3", fc1="h2w")),
    ("h3, fc2 h2a", scheme("h3", fc2="h2a")),
    ("h3, fc2 h2w", scheme("h3", fc2="h2w")),
    ("h3, last layer's qkv/qk/pv h1", scheme("h3", qkv_at_11="h1", qk_at_11="h1", pv_at_11="h1")),
    ("h3, linears h2a (weights single fp16)", scheme("h3", patch="h2a", qkv="h2a", out="h2a", fc1="h2a", fc2="h2a")),
    ("h3, patch h1 + pv h2w + qk h2a", scheme("h3", patch="h1", pv="h2w", qk="h2a")),
    ("h3, out + fc2 h2a (weights single fp16 in the two residual-writing linears)", scheme("h3", out="h2a", fc2="h2a")),
    ("h3, out + fc2 + patch h2a", scheme("h3", out="h2a", fc2="h2a", patch="h2a")),
    # cross terms on the double-rate e4m3 MFMA (linears only; attention keeps the 3-term fp16 form)
    ("mx: linears m2 (hh + 2 e4m3 cross terms)", scheme("h3", patch="m2", qkv="m2", out="m2", fc1="m2", fc2="m2")),
    ("mx: linears m2t (per-tensor scales)", scheme("h3", patch="m2t", qkv="m2t", out="m2t", fc1="m2t", fc2="m2t")),
    ("mx: linears m25 (a_lo fp16, w_lo e4m3)", scheme("h3", patch="m25", qkv="m25", out="m25", fc1="m25", fc2="m25")),
    ("mx: linears m15 (a_lo e4m3, w_lo dropped)", scheme("h3", patch="m15", qkv="m15", out="m15", fc1="m15", fc2="m15")),
    ("mx: fc1 + fc2 m2, rest h3", scheme("h3", fc1="m2", fc2="m2")),
    ("mx: qkv + fc1 m2, rest h3", scheme("h3", qkv="m2", fc1="m2")),
    ("mx: everything m2 (attention too)", scheme("m2")),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", nargs="+", default=["c1_b2_n50", "refdefault_b2_n64", "scales3_b2_n40", "unaligned_b3_n50", "vitl_b2_n70",
                                                   "nocalib_b2_n30", "c2shape_b4_n500"])
    ap.add_argument("--stress", action="store_true", help="add the trained-like-statistics weights (tests/helpers.stress_state)")
    ap.add_argument("--only", nargs="+", default=None, help="substring filter on scheme names")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    work = []
    for name in a.cases:
        g, kw, spec, sd, (patches, pos, scales) = helpers.load_case(name)
        work.append((name, spec, O.to_torch(sd), helpers.split_inputs(patches, pos, scales)))
    if a.stress:
        g, kw, spec, sd, (patches, pos, scales) = helpers.load_case("c1_b2_n50")
        for qk in (3.0, 5.0):
            work.append((f"stress_qk{qk:g}", spec, O.to_torch(helpers.stress_state(spec, 5, qk=qk)), helpers.split_inputs(patches, pos, scales)))
    # reference = the oracle in fp64 (the "exact" row then shows the fp32 noise floor of everything that is not a contraction)
    refs = {}
    for name, spec, sd, inp in work:
        sd64 = {k: v.double() for k, v in sd.items()}
        inp64 = tuple(tuple(None if t is None else t.double() for t in grp) for grp in inp)
        refs[name] = O.vtamiq_forward(sd64, spec, *inp64)[0].numpy()
    spec0 = work[0][1]
    wts = flops_weights(spec0, 500)
    print("# per case: max |q - q_ref| / |q_ref| over the scores with |q_ref| >= 0.1 rms(q_ref)  /  max |q - q_ref| / rms(q_ref) over all scores;")
    print("# q_ref = the oracle in fp64.  mfma/prod = MFMAs per product averaged over the forward's flops (ViT-B/16, N=500).")
    print("scheme".ljust(46), "mfma/prod", *[n[:17].rjust(19) for n, *_ in work], "  max raw   max nrm")
    for sname, sch in SCHEMES:
        if a.only and not any(s in sname for s in a.only):
            continue
        avg = sum(wts[k] * MFMAS[sch[k]] for k in SITES)
        raws, nrms = [], []
        for name, spec, sd, inp in work:
            q = forward(sd, spec, inp, sch)
            ref = refs[name]
            rms = float(np.sqrt(np.mean(ref ** 2)))
            d = np.abs(q - ref)
            ok = np.abs(ref) >= 0.1 * rms
            raws.append(float(np.max(d[ok] / np.abs(ref[ok]))))
            nrms.append(float(d.max() / rms))
        print(sname.ljust(46), f"{avg:9.2f}", *[f"{r:9.1e}/{n:9.1e}" for r, n in zip(raws, nrms)], f"{max(raws):9.1e} {max(nrms):9.1e}", flush=True)


if __name__ == "__main__":
    main()

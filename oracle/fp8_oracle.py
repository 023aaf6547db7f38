"""ORACLE -- test infrastructure only.  NOT part of the product path.

Fake-quant CPU restatement of the engine's "fp8" numerics mode (include/vtamiq_hip.h VTQ_PREC_FP8, BASELINE.json configs[4]):
the reference model of oracle/vtamiq_oracle.py with every encoder / patch-embedding linear layer evaluated on OCP e4m3
operands -- the operands are rounded to e4m3 exactly where the HIP path rounds them, the products are exact and the sums are
fp32 (fp64 on request), which is what the MX-scaled MFMA with unit block scales computes.  It answers "does the HIP fp8
path compute the fp8 model it claims to", NOT "is the fp8 model within 1e-3 of the fp32 reference" (it is not; DESIGN.md
section 2 quotes both distances).

Parity status: the fp32 restatement this file wraps is PINNED (tests/test_oracle_golden.py); the fp8 rounding points are this
repo's own definition (there is no fp8 path in the reference), checked for self-consistency in tests/test_fp8_oracle.py.

Quantisation points:
  * weights of patch_embeddings, query/key/value, attn.out, ffn.fc1, ffn.fc2: per OUTPUT channel n, s_n = the largest power of
    two with max|W[n, :]| * s_n <= 448;  W8 = e4m3(W * s_n);  the product is multiplied by 1 / s_n afterwards;
  * activations, per-tensor power-of-two scales, one per point (`Scales`): the flattened patches, and per layer the LayerNorm-1
    output, the attention context, the LayerNorm-2 output and the GELU output.  STATIC (256, 8, 16, 8, 4: the engine's defaults
    before calibration, vtamiq_amd/csrc/engine.hip kSPatch ...) or CALIBRATED as the engine does it on its first batch:
    `calibrate` runs this model on that batch and takes, point by point, the largest power of two that maps max |value| to
    <= 224 (pick_scale = engine.hip fp8_pick_scale); the tests feed the oracle the engine's own scales (vtq_fp8_get_scales).
    Values are clamped to +-448 before rounding (RNE, subnormals kept: torch's float8_e4m3fn conversion);
  * everything else (embedding sums, softmax, residual stream, LayerNorm statistics, biases, DiffNet head) is as in the fp32 oracle.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import vtamiq_oracle as vo

Tensor = torch.Tensor
E4M3_MAX = 448.0
S_PATCH, S_LN, S_ATT, S_GELU = 256.0, 8.0, 16.0, 4.0         # the static defaults
FP8_TARGET = 224.0                                            # calibration maps max |value| to <= this (engine.hip kFp8Target)


class Scales:
    """Activation scales of the fp8 model: patch, and per layer ln1 / att / ln2 / gelu."""

    def __init__(self, num_layers: int, patch=S_PATCH, ln1=None, att=None, ln2=None, gelu=None):
        self.patch = float(patch)
        self.ln1 = list(ln1) if ln1 is not None else [S_LN] * num_layers
        self.att = list(att) if att is not None else [S_ATT] * num_layers
        self.ln2 = list(ln2) if ln2 is not None else [S_LN] * num_layers
        self.gelu = list(gelu) if gelu is not None else [S_GELU] * num_layers

    @classmethod
    def from_engine(cls, d: dict) -> "Scales":
        """From vtamiq_amd.VTAMIQ.fp8_scales() (= vtq_fp8_get_scales)."""
        return cls(len(d["ln1"]), d["patch"], d["ln1"], d["att"], d["ln2"], d["gelu"])


def pick_scale(amax: float, keep: float) -> float:
    """Largest power of two s with amax * s <= FP8_TARGET, exactly (frexp); amax <= 0 or non-finite: keep."""
    if not (amax > 0.0) or not math.isfinite(amax):
        return keep
    f, e = math.frexp(amax)                                   # amax = f * 2^e, f in [0.5, 1);  224 = 0.875 * 2^8
    return math.ldexp(1.0, (8 if f <= 0.875 else 7) - e)


def to_e4m3(x: Tensor) -> Tensor:
    """Round to the e4m3 grid (clamp to +-448 first), returned in x's dtype."""
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float32).to(torch.float8_e4m3fn).to(x.dtype)


def quant_act(x: Tensor, scale: float) -> Tensor:
    """e4m3(x * scale): the operand bytes' values (still scaled)."""
    return to_e4m3(x * scale)


def row_scales(W: Tensor) -> Tensor:
    """Largest power of two s with max|row| * s <= 448 (1 for an all-zero row), exactly (frexp, no log)."""
    m = W.abs().amax(dim=1)
    f, e = torch.frexp(m.to(torch.float64))                  # m = f * 2^e, f in [0.5, 1);  448 = 0.875 * 2^9
    k = torch.where(f <= 0.875, 9 - e, 8 - e)
    s = torch.ldexp(torch.ones_like(f), k)
    return torch.where(m > 0, s, torch.ones_like(s)).to(W.dtype)


def quant_rows(W: Tensor) -> Tuple[Tensor, Tensor]:
    """(e4m3(W * s_n) values, 1 / s_n)."""
    s = row_scales(W)
    return to_e4m3(W * s[:, None]), 1.0 / s


def linear8(a8: Tensor, a_scale: float, W: Tensor, b: Tensor) -> Tensor:
    """(a8 @ W8^T) * (1 / s_n) * (1 / a_scale) + b with a8 already on the e4m3 grid."""
    W8, inv = quant_rows(W)
    return (a8 @ W8.t()) * (inv * (1.0 / a_scale)) + b


def embeddings(sd: Dict[str, Tensor], spec, patches: Tensor, pos: Tensor, scales: Optional[Tensor], s8: Optional[Scales] = None,
               calib: Optional[dict] = None) -> Tensor:
    """vtamiq_oracle.embeddings with the patch projection on e4m3 operands."""
    s8 = s8 or Scales(spec.num_layers)
    B, N = patches.shape[:2]
    e = "transformer.embeddings."
    W = sd[e + "patch_embeddings.weight"]
    H = W.shape[0]
    flat = patches.reshape(B * N, -1)
    if calib is not None:
        calib["patch"] = max(calib.get("patch", 0.0), float(flat.abs().max()))
        if calib.get("apply"):
            s8.patch = pick_scale(calib["patch"], s8.patch)
    a8 = quant_act(flat, s8.patch)
    x = linear8(a8, s8.patch, W.reshape(H, -1), sd[e + "patch_embeddings.bias"]).view(B, N, H)
    use_pos = getattr(spec, "use_pos_embedding", True)
    if use_pos:
        table = sd[e + "positional_embeddings.positional_embeddings"][0]
        x = x + table[vo.pos_index(pos.reshape(B * N, 2), spec.pos_grid)].view(B, N, -1)
    if spec.use_scale_embedding:
        if scales is None:
            raise ValueError("Model uses scale embedding but scales is passed as None.")
        st = sd[e + "scale_embeddings.scale_embeddings"][0]
        x = x + st[vo.scale_index(scales.reshape(B * N), spec.num_scales)].view(B, N, -1)
    cls = sd[e + "cls_token"].expand(B, 1, -1)
    if use_pos:
        cls = cls + table[0]
    toks = [cls]
    if spec.num_extra_tokens > 0:
        toks.append(sd[e + "extra_tokens"].expand(B, spec.num_extra_tokens, -1))
    return torch.cat(toks + [x], dim=1)


def encoder_layer(sd: Dict[str, Tensor], spec, i: int, x: Tensor, s8: Optional[Scales] = None, pick=None) -> Tensor:
    """pick(name, i, tensor): calibration hook, called with every pre-quantisation tensor before its scale is used."""
    s8 = s8 or Scales(spec.num_layers)
    pick = pick or (lambda name, i, t: None)
    p = f"transformer.encoder.layers.{i}."
    B, S, H = x.shape
    nh, dh = spec.num_heads, H // spec.num_heads
    ln = vo._layer_norm(x, sd[p + "attention_norm.weight"], sd[p + "attention_norm.bias"])
    pick("ln1", i, ln)
    ln8 = quant_act(ln, s8.ln1[i])

    def proj(nm):
        y = linear8(ln8, s8.ln1[i], sd[f"{p}attn.{nm}.weight"], sd[f"{p}attn.{nm}.bias"])
        return y.view(B, S, nh, dh).permute(0, 2, 1, 3)

    q, k, v = proj("query"), proj("key"), proj("value")
    probs = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, S, H)
    pick("att", i, ctx)
    ctx8 = quant_act(ctx, s8.att[i])
    h = linear8(ctx8, s8.att[i], sd[p + "attn.out.weight"], sd[p + "attn.out.bias"])
    if spec.use_layer_scale:
        h = h * sd[p + "ls1.gamma"]
    x = x + h
    ln = vo._layer_norm(x, sd[p + "ffn_norm.weight"], sd[p + "ffn_norm.bias"])
    pick("ln2", i, ln)
    ln8 = quant_act(ln, s8.ln2[i])
    gl = F.gelu(linear8(ln8, s8.ln2[i], sd[p + "ffn.fc1.weight"], sd[p + "ffn.fc1.bias"]))
    pick("gelu", i, gl)
    g8 = quant_act(gl, s8.gelu[i])
    h = linear8(g8, s8.gelu[i], sd[p + "ffn.fc2.weight"], sd[p + "ffn.fc2.bias"])
    if spec.use_layer_scale:
        h = h * sd[p + "ls2.gamma"]
    return x + h


def vit_tokens(sd, spec, patches, pos, scales, trace: Optional[List[Tensor]] = None, s8: Optional[Scales] = None) -> Tensor:
    x = embeddings(sd, spec, patches, pos, scales, s8)
    if trace is not None:
        trace.append(x[:, :spec.num_tokens].clone())
    for i in range(spec.num_layers):
        x = encoder_layer(sd, spec, i, x, s8)
        if trace is not None:
            trace.append(x[:, :spec.num_tokens].clone())
    x = vo._layer_norm(x, sd["transformer.encoder.encoder_norm.weight"], sd["transformer.encoder.encoder_norm.bias"])
    return x[:, :spec.num_tokens]


@torch.no_grad()
def calibrate(sd, spec, patches: Sequence[Tensor], pos: Sequence[Tensor], scales: Sequence[Optional[Tensor]]) -> Scales:
    """The engine's calibration forward restated: both images of the batch go through the encoder together; at every
    quantisation point the scale becomes pick_scale(max |value| over the whole batch) BEFORE the values are rounded with it."""
    s8 = Scales(spec.num_layers)
    cat = lambda ts: None if ts[0] is None else torch.cat(list(ts), dim=0)
    x = embeddings(sd, spec, cat(patches), cat(pos), cat(scales), s8, calib={"apply": True})

    def pick(name, i, t):
        cur = getattr(s8, name)
        cur[i] = pick_scale(float(t.abs().max()), cur[i])
    for i in range(spec.num_layers):
        x = encoder_layer(sd, spec, i, x, s8, pick)
    return s8


@torch.no_grad()
def vtamiq_forward(sd, spec, patches: Sequence[Tensor], pos: Sequence[Tensor], scales: Sequence[Optional[Tensor]],
                   trace: Optional[dict] = None, s8: Optional[Scales] = None):
    """The fp8 model's VTAMIQ.forward: (q, None).  trace["tokens"]: (L+1, 2B, T, H) pre-norm token rows, ref then dist.
    s8: activation scales (None: the static defaults)."""
    tr_r = [] if trace is not None else None
    tr_d = [] if trace is not None else None
    t_ref = vit_tokens(sd, spec, patches[0], pos[0], scales[0], tr_r, s8)
    t_dist = vit_tokens(sd, spec, patches[1], pos[1], scales[1], tr_d, s8)
    if trace is not None:
        trace["tokens"] = torch.cat([torch.stack(tr_r), torch.stack(tr_d)], dim=1)
    return vo.head(sd, spec, t_ref, t_dist), None

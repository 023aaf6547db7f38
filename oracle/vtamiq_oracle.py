"""ORACLE -- test infrastructure only.  NOT part of the product path.

CPU restatement (plain torch tensor ops, fp32 by default, fp64 on request) of the reference's
VTAMIQ pair-forward path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; `vtamiq_amd/` never does (tests/test_layout.py enforces it).

Parity status: PINNED.  The reference has no tests or golden vectors of its own,
so this restatement is pinned against outputs of the reference itself, imported in the build container
with a `timm` stub (tests/golden/make_golden.py; fixtures tests/golden/*.npz) -- see
tests/test_oracle_golden.py.

Each function cites the reference lines it restates (paths relative to the reference root).
Weights are passed as a flat dict {state_dict key: tensor} using the reference's key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    # torch.nn.LayerNorm(hidden, eps=1e-6): transformer.py:253-254, :350
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-6)


def patch_embed(sd: Dict[str, Tensor], patches: Tensor) -> Tensor:
    """Conv2d(3,H,k=P,s=P) on (B*N,3,P,P) == GEMM over the c-major flattened patch.  transformer.py:475-480, 527-536."""
    B, N = patches.shape[:2]
    W = sd["transformer.embeddings.patch_embeddings.weight"]
    b = sd["transformer.embeddings.patch_embeddings.bias"]
    H = W.shape[0]
    a = patches.reshape(B * N, -1)
    return (a @ W.reshape(H, -1).t() + b).view(B, N, H)


def pos_index(pos: Tensor, grid: int) -> Tensor:
    """floor(pos*G) -> row*G + col + 1, computed in the input float dtype then cast.  transformer.py:417-423."""
    p = torch.floor(pos * grid)
    idx = p[..., 0] * grid + p[..., 1] + 1
    return idx.to(torch.long)


def scale_index(scales: Tensor, num_scales: int) -> Tensor:
    """clamp(scale, 0, num_scales-1) + 1.  transformer.py:396-400."""
    return (torch.clamp(scales, 0, num_scales - 1) + 1).to(torch.long)


def embeddings(sd: Dict[str, Tensor], spec, patches: Tensor, pos: Tensor, scales: Optional[Tensor]) -> Tensor:
    """Embeddings.forward + forward_tokens.  transformer.py:507-562."""
    B, N = patches.shape[:2]
    e = "transformer.embeddings."
    x = patch_embed(sd, patches) if patches.dim() == 5 else patches           # (B, N, H) input skips the convolution (transformer.py:527-535)
    use_pos = getattr(spec, "use_pos_embedding", True)
    if use_pos:                                                               # transformer.py:539-543
        table = sd[e + "positional_embeddings.positional_embeddings"][0]      # (G*G+1, H)
        x = x + table[pos_index(pos.reshape(B * N, 2), spec.pos_grid)].view(B, N, -1)
    if spec.use_scale_embedding:
        if scales is None:
            raise ValueError("Model uses scale embedding but scales is passed as None.")   # transformer.py:547-548
        st = sd[e + "scale_embeddings.scale_embeddings"][0]
        x = x + st[scale_index(scales.reshape(B * N), spec.num_scales)].view(B, N, -1)
    cls = sd[e + "cls_token"].expand(B, 1, -1)
    if use_pos:
        cls = cls + table[0]                                                  # CLS gets pos row 0 (:511-516)
    toks = [cls]
    if spec.num_extra_tokens > 0:
        toks.append(sd[e + "extra_tokens"].expand(B, spec.num_extra_tokens, -1))   # no pos/scale (:520-523)
    return torch.cat(toks + [x], dim=1)


def attention(sd: Dict[str, Tensor], prefix: str, x: Tensor, num_heads: int, return_probs: bool = False):
    """MultiHeadSelfAttention.forward.  transformer.py:153-172."""
    B, S, H = x.shape
    dh = H // num_heads

    def proj(nm):
        y = x @ sd[f"{prefix}attn.{nm}.weight"].t() + sd[f"{prefix}attn.{nm}.bias"]
        return y.view(B, S, num_heads, dh).permute(0, 2, 1, 3)

    q, k, v = proj("query"), proj("key"), proj("value")
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    probs = torch.softmax(scores, dim=-1)
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, S, H)
    out = ctx @ sd[f"{prefix}attn.out.weight"].t() + sd[f"{prefix}attn.out.bias"]
    return (out, probs) if return_probs else out


def mlp(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """MLP.forward with exact-erf GELU.  transformer.py:212-215, 54-57."""
    h = F.gelu(x @ sd[prefix + "ffn.fc1.weight"].t() + sd[prefix + "ffn.fc1.bias"])
    return h @ sd[prefix + "ffn.fc2.weight"].t() + sd[prefix + "ffn.fc2.bias"]


def adapter(sd: Dict[str, Tensor], prefix: str, h: Tensor) -> Tensor:
    """Adapter.forward: h + Linear(gelu(Linear(h))).  transformer.py:177-194 (nn.GELU() = exact erf form)."""
    a = F.gelu(h @ sd[prefix + "adapter.0.weight"].t() + sd[prefix + "adapter.0.bias"])
    return h + a @ sd[prefix + "adapter.2.weight"].t() + sd[prefix + "adapter.2.bias"]


def encoder_layer(sd: Dict[str, Tensor], spec, i: int, x: Tensor) -> Tensor:
    """EncoderLayer.forward (pre-LN; DropPath is identity).  transformer.py:275-285.
    With num_adapters > 0 the forward applies adapter pair 0 (backbone.py:54-57: adapter_num defaults to 0)."""
    p = f"transformer.encoder.layers.{i}."
    ad = getattr(spec, "num_adapters", 0) > 0
    h = attention(sd, p, _layer_norm(x, sd[p + "attention_norm.weight"], sd[p + "attention_norm.bias"]), spec.num_heads)
    if ad:
        h = adapter(sd, p + "adapter1.", h)
    if spec.use_layer_scale:
        h = h * sd[p + "ls1.gamma"]
    x = x + h
    h = mlp(sd, p, _layer_norm(x, sd[p + "ffn_norm.weight"], sd[p + "ffn_norm.bias"]))
    if ad:
        h = adapter(sd, p + "adapter2.", h)
    if spec.use_layer_scale:
        h = h * sd[p + "ls2.gamma"]
    return x + h


def vit_tokens(sd: Dict[str, Tensor], spec, patches: Tensor, pos: Tensor, scales: Optional[Tensor],
               trace: Optional[List[Tensor]] = None) -> Tensor:
    """VisionTransformer.forward(tokens_only=True): embeddings -> L layers -> encoder_norm -> x[:, :T].
    transformer.py:628-641, 363-378.  `trace` collects the pre-norm token rows after every layer."""
    x = embeddings(sd, spec, patches, pos, scales)
    T = spec.num_tokens
    if trace is not None:
        trace.append(x[:, :T].clone())
    for i in range(spec.num_layers):
        x = encoder_layer(sd, spec, i, x)
        if trace is not None:
            trace.append(x[:, :T].clone())
    x = _layer_norm(x, sd["transformer.encoder.encoder_norm.weight"], sd["transformer.encoder.encoder_norm.bias"])
    return x[:, :T]


def _prelu(x: Tensor, a: Tensor) -> Tensor:
    return torch.where(x >= 0, x, a * x)        # nn.PReLU() with one shared slope (channel_attention.py:43)


def _conv1x1(sd, key: str, x: Tensor) -> Tensor:
    w = sd[key + ".weight"]
    return x @ w.reshape(w.shape[0], w.shape[1]).t() + sd[key + ".bias"]   # Conv1d(k=1) on (B,C,1) == Linear on (B,C)


def rcab(sd, prefix: str, x: Tensor) -> Tensor:
    """RCAB: x + CA(Conv(PReLU(x))).  channel_attention.py:41-50, 77-86 (AdaptiveAvgPool1d(1) on length 1 = identity)."""
    c = _conv1x1(sd, prefix + "body.2", _prelu(x, sd[prefix + "body.1.weight"]))
    t = torch.relu(_conv1x1(sd, prefix + "body.4.conv_du.1", c))
    w = torch.sigmoid(_conv1x1(sd, prefix + "body.4.conv_du.4", t))
    return x + c * w


def quality_decoder(sd, spec, d: Tensor) -> Tensor:
    """get_quality_decoder: RG x num_rgs then Conv1d.  vtamiq.py:12-23; ResidualGroup channel_attention.py:13-29
    (DropPath identity in eval)."""
    if not spec.calibrate:
        return d
    x = d
    for g in range(spec.num_rgs):
        y = x
        for k in range(spec.num_rcabs):
            y = rcab(sd, f"quality_decoder.{g}.body.{k}.", y)
        x = x + _conv1x1(sd, f"quality_decoder.{g}.body.{spec.num_rcabs}", y)
    return _conv1x1(sd, f"quality_decoder.{spec.num_rgs}", x)


def q_predictor(sd, x: Tensor) -> Tensor:
    """Dropout(eval: identity) -> Linear(H,H/4) -> PReLU -> Dropout -> Linear(H/4,1) -> flatten.  vtamiq.py:71-77,116-117."""
    h = x @ sd["q_predictor.1.weight"].t() + sd["q_predictor.1.bias"]
    h = _prelu(h, sd["q_predictor.2.weight"])
    return (h @ sd["q_predictor.4.weight"].t() + sd["q_predictor.4.bias"]).flatten()


def head(sd, spec, tok_ref: Tensor, tok_dist: Tensor, token_num: int = 0) -> Tensor:
    """vtamiq.py:104-117 from the (B,T,H) token rows: token `token_num` (vtamiq.py:57: 0) diff -> diff_scale -> quality_decoder -> q_predictor."""
    d = tok_ref[:, token_num] - tok_dist[:, token_num]
    if spec.diff_scale:
        d = d * sd["diff_scale.gamma"]
    return q_predictor(sd, quality_decoder(sd, spec, d))


@torch.no_grad()
def vtamiq_forward(sd: Dict[str, Tensor], spec, patches: Sequence[Tensor], pos: Sequence[Tensor],
                   scales: Sequence[Optional[Tensor]], trace: Optional[dict] = None, token_num: int = 0) -> Tuple[Tensor, None]:
    """VTAMIQ.forward (eval mode).  vtamiq.py:94-119.  Returns (q, None) like the reference."""
    tr_r = [] if trace is not None else None
    tr_d = [] if trace is not None else None
    t_ref = vit_tokens(sd, spec, patches[0], pos[0], scales[0], tr_r)     # two serial passes, vtamiq.py:100-101
    t_dist = vit_tokens(sd, spec, patches[1], pos[1], scales[1], tr_d)
    if trace is not None:
        trace["tokens_ref"], trace["tokens_dist"] = torch.stack(tr_r), torch.stack(tr_d)
    return head(sd, spec, t_ref, t_dist, token_num), None


def predict(sd, spec, batch, use_scales: Optional[bool] = None, dtype=torch.float32):
    """train.get_data_tuple + train.predict (non-pairwise branch).  train.py:254-255, 258-267, 302-314.

    batch = (q[B], patches[B,2,N,3,P,P], pos[B,2,N,2], scales[B,2,N] | scales[B] == -1).  Every element is
    cast to `dtype` first (the reference casts scales to float32 too), then split per image along axis 1.
    """
    q, patches, pos, scales = (torch.as_tensor(t).to(dtype) for t in batch[:4])
    if use_scales is None:
        use_scales = spec.use_scale_embedding
    p = tuple(patches[:, i].clone() for i in range(patches.shape[1]))
    ps = tuple(pos[:, i].clone() for i in range(pos.shape[1]))
    sc = tuple(scales[:, i].clone() for i in range(scales.shape[1])) if use_scales else (None, None)
    q_p = vtamiq_forward(sd, spec, p, ps, sc)[0].flatten()
    return q, q_p


def to_torch(sd_np, dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.as_tensor(v).to(dtype) for k, v in sd_np.items()}

"""Model topology for the VTAMIQ pair-forward path.

Mirrors the reference's configuration surface for this path:
  * ViT variants          -- modules/VisionTransformer/transformer.py:68-111
  * backbone kwargs       -- modules/VisionTransformer/backbone.py:17-34
  * VTAMIQ ctor kwargs    -- modules/vtamiq/vtamiq.py:27-46
  * config-file defaults  -- train_config.py:169-194

`ModelSpec` is pure host-side bookkeeping: it fixes every shape the HIP engine,
the synthetic generator and the oracle need, and it enumerates the reference's
state_dict key layout.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import List, Tuple

VIT_VARIANT_B8 = "ViT-B8"
VIT_VARIANT_B16 = "ViT-B16"
VIT_VARIANT_L16 = "ViT-L16"

# transformer.py:68-98 (img_dim, patch_size, hidden, mlp, heads, layers)
_VIT_VARIANTS = {
    VIT_VARIANT_B16: dict(img_dim=384, patch_size=16, hidden_size=768, mlp_dim=3072, num_heads=12, num_layers=12),
    VIT_VARIANT_B8: dict(img_dim=384, patch_size=8, hidden_size=768, mlp_dim=3072, num_heads=12, num_layers=12),
    VIT_VARIANT_L16: dict(img_dim=384, patch_size=16, hidden_size=1024, mlp_dim=4096, num_heads=16, num_layers=24),
}


def get_vit_config(variant: str) -> dict:
    """Same contract as transformer.py:101-111 (ValueError on unknown variant)."""
    if variant not in _VIT_VARIANTS:
        raise ValueError("ViT: Unsupported variant [{}], pick from {}.".format(
            variant, [VIT_VARIANT_B8, VIT_VARIANT_B16, VIT_VARIANT_L16]))
    return dict(_VIT_VARIANTS[variant])


@dataclass(frozen=True)
class ModelSpec:
    # ViT
    variant: str = VIT_VARIANT_B16
    hidden_size: int = 768
    mlp_dim: int = 3072
    num_heads: int = 12
    num_layers: int = 12          # layers actually kept (num_keep_layers applied, transformer.py:342-345)
    patch_size: int = 16
    pos_grid: int = 24            # img_dim // patch_size (transformer.py:411)
    num_extra_tokens: int = 0
    num_scales: int = 0           # scale embedding active iff num_scales > 1 (transformer.py:500)
    use_layer_scale: bool = False
    num_adapters: int = 0         # Adapter pairs per layer (transformer.py:177-194, 260-269); the forward uses pair 0 (backbone.py:54-57)
    use_patch_embedding: bool = True # False: no patch Conv2d; the model is fed pre-embedded (B, N, H) rows (transformer.py:473-480, 534-535)
    use_pos_embedding: bool = True   # False: no UvPosEmbedding module, nothing added to the patches or to CLS (transformer.py:497-499, 514, 539)
    # VTAMIQ head
    calibrate: bool = True
    diff_scale: bool = True
    num_rgs: int = 4
    num_rcabs: int = 4
    ca_reduction: int = 8

    @property
    def num_tokens(self) -> int:          # CLS + registers (transformer.py:497)
        return 1 + self.num_extra_tokens

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:           # 3 * P * P, the K of the patch-embedding GEMM
        return 3 * self.patch_size * self.patch_size

    @property
    def use_scale_embedding(self) -> bool:
        return self.num_scales > 1

    @property
    def ca_hidden(self) -> int:           # channel_attention.py:75
        return self.hidden_size // self.ca_reduction

    @property
    def pred_hidden(self) -> int:         # vtamiq.py:73
        return self.hidden_size // 4

    def seq_len(self, num_patches: int) -> int:
        return num_patches + self.num_tokens

    def asdict(self) -> dict:
        return asdict(self)

    # ---- algorithmic flop model, SURVEY.md section 8(d) -------------------------------------
    def flops_per_image(self, num_patches: int) -> float:
        H, M, L = self.hidden_size, self.mlp_dim, self.num_layers
        S = self.seq_len(num_patches)
        f = 2.0 * num_patches * self.patch_dim * H + L * (8.0 * S * H * H + 4.0 * S * H * M + 4.0 * S * S * H)
        if self.num_adapters > 0:                       # two bottleneck adapters per layer: H -> H/4 -> H each
            f += L * 2.0 * (4.0 * S * H * (H // 4))
        return f

    def flops_head(self) -> float:
        H = self.hidden_size
        f = 2.0 * H * (H // 4) + 2.0 * (H // 4)
        if self.calibrate:
            f += self.num_rgs * (self.num_rcabs * (2.0 * H * H + 4.0 * H * self.ca_hidden) + 2.0 * H * H) + 2.0 * H * H
        return f

    def flops_per_pair(self, num_patches: int) -> float:
        return 2.0 * self.flops_per_image(num_patches) + self.flops_head()

    def flops_per_pair_executed(self, num_patches: int, cls_prune: bool = True) -> float:
        """Dense flops the engine actually executes per pair.  With the CLS-only last layer (only token 0 is consumed,
        vtamiq.py:107-108) the last layer keeps the K/V projections for every row and runs Q, attention, out-proj and the
        MLP for one row per image."""
        if not cls_prune or self.num_adapters > 0:      # the engine runs the full last layer when adapters are on
            return self.flops_per_pair(num_patches)
        H, M = self.hidden_size, self.mlp_dim
        S = self.seq_len(num_patches)
        full_last = 8.0 * S * H * H + 4.0 * S * H * M + 4.0 * S * S * H
        pruned_last = 4.0 * S * H * H + (4.0 * H * H + 4.0 * H * M + 4.0 * S * H)
        return self.flops_per_pair(num_patches) - 2.0 * (full_last - pruned_last)

    # ---- state_dict layout (reference key names) -----------------------
    def state_layout(self) -> List[Tuple[str, Tuple[int, ...], str]]:
        """[(key, shape, kind)] in the reference's registration order.

        kind in {matrix, bias, embed, ln_w, ln_b, gamma, prelu} drives the synthetic generator only.
        """
        H, M, P = self.hidden_size, self.mlp_dim, self.patch_size
        out: List[Tuple[str, Tuple[int, ...], str]] = []
        e = "transformer.embeddings."
        out.append((e + "cls_token", (1, 1, H), "embed"))
        if self.num_extra_tokens > 0:
            out.append((e + "extra_tokens", (1, self.num_extra_tokens, H), "embed"))
        if self.use_patch_embedding:
            out.append((e + "patch_embeddings.weight", (H, 3, P, P), "matrix"))
            out.append((e + "patch_embeddings.bias", (H,), "bias"))
        if self.use_pos_embedding:
            out.append((e + "positional_embeddings.positional_embeddings", (1, self.pos_grid ** 2 + 1, H), "embed"))
        if self.use_scale_embedding:
            out.append((e + "scale_embeddings.scale_embeddings", (1, self.num_scales + 1, H), "embed"))
        enc = "transformer.encoder."
        out.append((enc + "encoder_norm.weight", (H,), "ln_w"))
        out.append((enc + "encoder_norm.bias", (H,), "ln_b"))
        for i in range(self.num_layers):
            p = f"{enc}layers.{i}."
            out.append((p + "attention_norm.weight", (H,), "ln_w"))
            out.append((p + "attention_norm.bias", (H,), "ln_b"))
            out.append((p + "ffn_norm.weight", (H,), "ln_w"))
            out.append((p + "ffn_norm.bias", (H,), "ln_b"))
            out.append((p + "ffn.fc1.weight", (M, H), "matrix"))
            out.append((p + "ffn.fc1.bias", (M,), "bias"))
            out.append((p + "ffn.fc2.weight", (H, M), "matrix"))
            out.append((p + "ffn.fc2.bias", (H,), "bias"))
            for nm in ("query", "key", "value", "out"):
                out.append((p + f"attn.{nm}.weight", (H, H), "matrix"))
                out.append((p + f"attn.{nm}.bias", (H,), "bias"))
            for a in range(1, 2 * self.num_adapters + 1):           # adapter{2j+1} (after attention), adapter{2j+2} (after the MLP)
                q = f"{p}adapter{a}.adapter."
                out.append((q + "0.weight", (H // 4, H), "matrix"))
                out.append((q + "0.bias", (H // 4,), "bias"))
                out.append((q + "2.weight", (H, H // 4), "matrix"))
                out.append((q + "2.bias", (H,), "bias"))
            if self.use_layer_scale:
                out.append((p + "ls1.gamma", (H,), "gamma"))
                out.append((p + "ls2.gamma", (H,), "gamma"))
        if self.diff_scale:
            out.append(("diff_scale.gamma", (H,), "gamma"))
        if self.calibrate:
            hid = self.ca_hidden
            for g in range(self.num_rgs):
                for k in range(self.num_rcabs):
                    p = f"quality_decoder.{g}.body.{k}.body."
                    out.append((p + "1.weight", (1,), "prelu"))
                    out.append((p + "2.weight", (H, H, 1), "matrix"))
                    out.append((p + "2.bias", (H,), "bias"))
                    out.append((p + "4.conv_du.1.weight", (hid, H, 1), "matrix"))
                    out.append((p + "4.conv_du.1.bias", (hid,), "bias"))
                    out.append((p + "4.conv_du.4.weight", (H, hid, 1), "matrix"))
                    out.append((p + "4.conv_du.4.bias", (H,), "bias"))
                p = f"quality_decoder.{g}.body.{self.num_rcabs}."
                out.append((p + "weight", (H, H, 1), "matrix"))
                out.append((p + "bias", (H,), "bias"))
            out.append((f"quality_decoder.{self.num_rgs}.weight", (H, H, 1), "matrix"))
            out.append((f"quality_decoder.{self.num_rgs}.bias", (H,), "bias"))
        out.append(("q_predictor.1.weight", (H // 4, H), "matrix"))
        out.append(("q_predictor.1.bias", (H // 4,), "bias"))
        out.append(("q_predictor.2.weight", (1,), "prelu"))
        out.append(("q_predictor.4.weight", (1, H // 4), "matrix"))
        out.append(("q_predictor.4.bias", (1,), "bias"))
        return out


def make_spec(vit_config: dict | None = None, *, calibrate=True, diff_scale=True, num_rgs=4, num_rcabs=4,
              ca_reduction=8, **_ignored) -> ModelSpec:
    """Build a ModelSpec from the reference's VTAMIQ ctor kwargs (vtamiq.py:27-46, backbone.py:17-34).

    Raises for the options the reference itself cannot run or that are outside the accelerated path
    (SURVEY.md section 8a 'known quirks' and row a13).
    """
    vc = dict(vit_config or {})
    vc.pop("use_classifier", None)                      # vtamiq.py:50 -- forced False
    variant = vc.pop("variant", VIT_VARIANT_B16)
    cfg = get_vit_config(variant)
    if not vc.pop("use_cls_token", True):
        raise ValueError("use_cls_token=False is not runnable in the reference (transformer.py:485) and is rejected here")
    use_patch = bool(vc.pop("use_patch_embedding", True))
    use_pos = bool(vc.pop("use_pos_embedding", True))
    num_adapters = int(vc.pop("num_adapters", 0))
    if num_adapters < 0:
        raise ValueError("num_adapters must be >= 0")
    rl, ra = vc.pop("return_layers", False), vc.pop("return_attention", False)
    if rl or ra:
        # forward_vit would return per-layer states / (B, h, S, S) attention maps, which VTAMIQ.forward discards (vtamiq.py:100-101:
        # `feats, _, _ = ...`): the scores are unaffected, so the options are accepted and nothing is materialised
        import warnings
        warnings.warn("return_layers / return_attention have no effect on VTAMIQ.forward's outputs and are not materialised by the "
                      "HIP path (vtq_set_token_trace taps the token rows of every layer for debugging)")
    vc.pop("pretrained", None)                          # weights arrive through load_state_dict
    vc.pop("path_drop_prob", None)                      # encoder DropPath is identity in the reference (transformer.py:272-273)
    num_keep = vc.pop("num_keep_layers", -1)
    num_layers = cfg["num_layers"]
    if 0 < num_keep:
        num_layers = max(1, min(num_keep, cfg["num_layers"]))   # transformer.py:343-345
    spec = ModelSpec(
        variant=variant, hidden_size=cfg["hidden_size"], mlp_dim=cfg["mlp_dim"], num_heads=cfg["num_heads"],
        num_layers=num_layers, patch_size=cfg["patch_size"], pos_grid=cfg["img_dim"] // cfg["patch_size"],
        num_extra_tokens=int(vc.pop("num_extra_tokens", 0)), num_scales=int(vc.pop("num_scales", 0)),
        use_layer_scale=bool(vc.pop("use_layer_scale", False)), num_adapters=num_adapters, use_patch_embedding=use_patch, use_pos_embedding=use_pos,
        calibrate=bool(calibrate), diff_scale=bool(diff_scale), num_rgs=int(num_rgs), num_rcabs=int(num_rcabs),
        ca_reduction=int(ca_reduction))
    if vc:
        import warnings
        warnings.warn(f"VisionTransformerBackbone: unused kwargs {sorted(vc)}")   # backbone.py:35 only warns
    if spec.patch_size not in (8, 16):
        raise NotImplementedError("only 16x16 (ViT-B16 / ViT-L16) and 8x8 (ViT-B8) patches are on the accelerated path")
    return spec

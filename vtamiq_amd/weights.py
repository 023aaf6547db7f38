"""Weight ingestion for the engine-backed VTAMIQ -- host-side only.

  * JAX ViT ``.npz`` -> state_dict entries: the conversion of ``VisionTransformer.load_from``
    (modules/VisionTransformer/transformer.py:643-668), ``EncoderLayer.load_from`` (:287-325: kernels reshaped to
    (H, H) and transposed to torch ``Linear`` layout), ``UvPosEmbedding.load_from`` (:428-455: bilinear
    ``scipy.ndimage.zoom`` of the position grid when sizes differ) and ``np2th`` (:118-122: HWIO -> OIHW).
  * ``.pth`` checkpoints written by ``train.save_checkpoint`` (train.py:222-251): ``{"model_state_dict": ...}`` with the
    optional dropping of ``transformer.`` / head keys (train.py:157-179) and the strict -> non-strict fallback of
    ``modules/utils.load_model`` (:81-91).
"""
from __future__ import annotations

import warnings
from typing import Dict, Mapping, Optional

import numpy as np
import torch

MODEL_STATE_DICT = "model_state_dict"          # train_config.py:53

# transformer.py:68-98: where the reference looks for the JAX checkpoints (relative to the working directory)
_VIT_WEIGHTS = {"ViT-B16": "./modules/VisionTransformer/weights/imagenet21k+imagenet2012_ViT-B_16.npz",
                "ViT-B8": "./modules/VisionTransformer/weights/imagenet21k+imagenet2012_ViT-B_8.npz",
                "ViT-L16": "./modules/VisionTransformer/weights/imagenet21k+imagenet2012_ViT-L_16.npz"}


def default_vit_weights_path(variant: str) -> str:
    return _VIT_WEIGHTS[variant]

_ROOT = "Transformer/encoderblock_{}"
_ATT = "MultiHeadDotProductAttention_1"


def _np2th(w: np.ndarray) -> torch.Tensor:
    """transformer.py:118-122 -- 4-D kernels are HWIO in the JAX checkpoint, OIHW in torch."""
    if w.ndim == 4:
        w = w.transpose([3, 2, 0, 1])
    return torch.from_numpy(np.ascontiguousarray(w))


def resize_pos_embedding(posemb: np.ndarray, num_tokens_new: int) -> np.ndarray:
    """transformer.py:428-455 -- keep the CLS row, bilinearly zoom the (gs x gs) grid of the remaining rows."""
    if posemb.shape[1] == num_tokens_new:
        return posemb
    from scipy import ndimage
    tok, grid = posemb[:, :1], posemb[0, 1:]
    gs_old = int(np.sqrt(len(grid)))
    gs_new = int(np.sqrt(num_tokens_new - 1))
    grid = grid.reshape(gs_old, gs_old, -1)
    zoom = (gs_new / gs_old, gs_new / gs_old, 1)
    grid = ndimage.zoom(grid, zoom, order=1).reshape(1, gs_new * gs_new, -1)
    return np.concatenate([tok, grid], axis=1)


def convert_vit_npz(weights: Mapping[str, np.ndarray], hidden_size: int, num_layers: int, num_pos_tokens: int
                    ) -> Dict[str, torch.Tensor]:
    """JAX ViT arrays -> ``transformer.*`` state_dict entries of the reference layout.

    Only the first ``num_layers`` encoder blocks are read (num_keep_layers truncation, transformer.py:343-345).
    ``extra_tokens``, ``scale_embeddings``, LayerScale gammas and the whole head are NOT in a ViT checkpoint and keep
    their current values (as in the reference, transformer.py:650-654).
    """
    H = hidden_size
    sd: Dict[str, torch.Tensor] = {}
    e = "transformer.embeddings."
    sd[e + "patch_embeddings.weight"] = _np2th(np.asarray(weights["embedding/kernel"]))
    sd[e + "patch_embeddings.bias"] = _np2th(np.asarray(weights["embedding/bias"]))
    sd[e + "cls_token"] = _np2th(np.asarray(weights["cls"]))
    pos = resize_pos_embedding(np.asarray(weights["Transformer/posembed_input/pos_embedding"]), num_pos_tokens)
    sd[e + "positional_embeddings.positional_embeddings"] = _np2th(pos.astype(np.float32))
    enc = "transformer.encoder."
    sd[enc + "encoder_norm.weight"] = _np2th(np.asarray(weights["Transformer/encoder_norm/scale"]))
    sd[enc + "encoder_norm.bias"] = _np2th(np.asarray(weights["Transformer/encoder_norm/bias"]))
    for i in range(num_layers):
        r = _ROOT.format(i)
        p = f"{enc}layers.{i}."
        for jax_name, th_name in (("query", "query"), ("key", "key"), ("value", "value"), ("out", "out")):
            k = np.asarray(weights[f"{r}/{_ATT}/{jax_name}/kernel"])
            sd[p + f"attn.{th_name}.weight"] = _np2th(k).reshape(H, H).t().contiguous()
            sd[p + f"attn.{th_name}.bias"] = _np2th(np.asarray(weights[f"{r}/{_ATT}/{jax_name}/bias"])).reshape(-1)
        sd[p + "ffn.fc1.weight"] = _np2th(np.asarray(weights[f"{r}/MlpBlock_3/Dense_0/kernel"])).t().contiguous()
        sd[p + "ffn.fc2.weight"] = _np2th(np.asarray(weights[f"{r}/MlpBlock_3/Dense_1/kernel"])).t().contiguous()
        sd[p + "ffn.fc1.bias"] = _np2th(np.asarray(weights[f"{r}/MlpBlock_3/Dense_0/bias"]))
        sd[p + "ffn.fc2.bias"] = _np2th(np.asarray(weights[f"{r}/MlpBlock_3/Dense_1/bias"]))
        sd[p + "attention_norm.weight"] = _np2th(np.asarray(weights[f"{r}/LayerNorm_0/scale"]))
        sd[p + "attention_norm.bias"] = _np2th(np.asarray(weights[f"{r}/LayerNorm_0/bias"]))
        sd[p + "ffn_norm.weight"] = _np2th(np.asarray(weights[f"{r}/LayerNorm_2/scale"]))
        sd[p + "ffn_norm.bias"] = _np2th(np.asarray(weights[f"{r}/LayerNorm_2/bias"]))
    return {k: v.to(torch.float32) for k, v in sd.items()}


def load_vit_npz(model, npz) -> None:
    """Load a JAX ViT checkpoint (path or mapping) into an engine-backed VTAMIQ: what ``pretrained=True`` does in the
    reference constructor (transformer.py:621-624)."""
    weights = np.load(npz) if isinstance(npz, (str, bytes)) or hasattr(npz, "__fspath__") else npz
    spec = model.spec
    sd = convert_vit_npz(weights, spec.hidden_size, spec.num_layers, spec.pos_grid ** 2 + 1)
    if not spec.use_pos_embedding:                      # transformer.py:656-657: the table is read only into a model that has one
        sd.pop("transformer.embeddings.positional_embeddings.positional_embeddings")
    if not spec.use_patch_embedding:                    # transformer.py:643-651: the conv weights are read only into a model that has the conv
        sd.pop("transformer.embeddings.patch_embeddings.weight")
        sd.pop("transformer.embeddings.patch_embeddings.bias")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if unexpected:
        raise RuntimeError(f"unexpected keys from the ViT checkpoint: {unexpected[:4]}")
    model.refresh_weights()


def load_checkpoint(model, checkpoint, allow_vit: bool = True, allow_head: bool = True, map_location="cpu") -> Optional[dict]:
    """``train.get_model`` for a ``.pth`` written by the reference (train.py:157-179): pick ``model_state_dict``, optionally
    drop the transformer / head entries, then strict load with the reference's non-strict fallback."""
    ckpt = torch.load(checkpoint, map_location=map_location, weights_only=False) if not isinstance(checkpoint, Mapping) else checkpoint
    sd = dict(ckpt[MODEL_STATE_DICT])

    def pop(prefix):
        for k in list(sd):
            if prefix in k:
                sd.pop(k)
    if not allow_vit:
        pop("transformer.")
    if not allow_head:
        for pfx in ("calibration_diff.", "calibration_feat.", "q_predictor."):    # the prefixes train.py:172-174 drops
            pop(pfx)
    try:
        model.load_state_dict(sd)
    except RuntimeError as err:                                                    # modules/utils.py:84-91
        warnings.warn(f"{err}\nContinuing with partial load...")
        model.load_state_dict(sd, strict=False)
    model.refresh_weights()
    return ckpt

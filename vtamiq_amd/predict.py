"""Host-side mirror of the call boundary in the reference driver (train.py:254-314).

`get_data_tuple`, `split_per_image` and `predict` keep the reference's names, argument meaning and return values so
that a validation loop written against train.py works unchanged with the MI355X model callable.
"""
from __future__ import annotations

import torch
from torch import nn


class PreferenceModule(nn.Module):
    """modules/vtamiq/common.py:5-14 -- sigmoid(p * (q2 - q1)) for pairwise datasets (state_dict key: "p")."""

    def __init__(self, weight=1.):
        super().__init__()
        self.p = nn.Parameter(torch.as_tensor(weight, dtype=torch.float32).reshape(-1))

    def forward(self, q1, q2):
        return torch.sigmoid(self.p * (q2 - q1)).flatten()


def get_data_tuple(batch, device):
    """train.py:254-255 -- every batch element to `device` as float32 (scales included)."""
    return tuple(data.to(device, dtype=torch.float32, non_blocking=True) for data in batch)


def split_per_image(x, has_batch_dim=True, clone=True):
    """train.py:258-267 -- a stacked (B, K, ...) tensor (or (K, ...) without batch dimension, which gains a leading 1) as K
    per-image tensors; `clone` makes each one contiguous (the engine needs that, the reference's .view() calls too)."""
    axis = 1 if has_batch_dim else 0
    pieces = x.unbind(axis)
    if not has_batch_dim:
        pieces = tuple(t.unsqueeze(0) for t in pieces)
    return tuple(t.clone() for t in pieces) if clone else tuple(pieces)


def model_forward(model, model_name, patches, pos, scales):
    """train.py:270-275 -- dispatch by model name; only VTAMIQ exists on this path."""
    if "vtamiq" not in model_name.lower():
        raise ValueError(f"Unsupported model {model_name}")
    return model(patches, pos, scales)


def predict(model, pref_module, data, is_pairwise, output_feats, use_scales):
    """train.py:278-314.  `data` = (q, patches, pos, scales, ...) as produced by get_data_tuple; returns (q, q_p, feats) with
    q_p flattened.  FR items: q_p is the model's score.  Pairwise items (ref, dist1, dist2): q_p is the preference,
    pref_module(q1, q2) when a module is given, otherwise sigmoid(q1 - q2) -- the reference's two branches disagree in sign
    (train.py:298 vs :301, PreferenceModule computes sigmoid(p (q2 - q1))); that is reproduced, not fixed."""
    q_true, stacked_patches, stacked_pos, stacked_scales = data[:4]
    per_image = [split_per_image(stacked_patches), split_per_image(stacked_pos),
                 split_per_image(stacked_scales) if use_scales else None]
    if not is_pairwise:
        scales = per_image[2] if use_scales else (None, None)
        out = model(per_image[0], per_image[1], scales)
        q_p, feats = (out[0], out[1]) if output_feats else (out[0], None)
        return q_true, q_p.flatten(), feats

    triplet = lambda k: tuple(per_image[k]) if per_image[k] is not None else (None, None, None)      # noqa: E731
    (p_ref, p_d1, p_d2), (x_ref, x_d1, x_d2), (s_ref, s_d1, s_d2) = triplet(0), triplet(1), triplet(2)
    if hasattr(model, "forward_pairwise"):
        # same scores as the reference's two model calls (train.py:286-287), the shared reference image encoded once
        q1, q2 = model.forward_pairwise((p_ref, p_d1, p_d2), (x_ref, x_d1, x_d2), (s_ref, s_d1, s_d2) if use_scales else None)
        feats = (None, None) if output_feats else None
    else:
        first = model((p_ref, p_d1), (x_ref, x_d1), (s_ref, s_d1))
        second = model((p_ref, p_d2), (x_ref, x_d2), (s_ref, s_d2))
        q1, q2 = first[0], second[0]
        feats = (first[1], second[1]) if output_feats else None
    q_p = pref_module(q1, q2) if pref_module is not None else torch.sigmoid(q1 - q2)
    return q_true, q_p.flatten(), feats

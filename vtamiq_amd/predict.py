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
    """train.py:258-267 -- (B,K,...) -> K tensors (B,...), cloned to contiguous."""
    num_images = x.shape[1] if has_batch_dim else x.shape[0]

    def x_i(i):
        return x[:, i] if has_batch_dim else x[i].unsqueeze(0)
    return tuple((x_i(i).clone() if clone else x_i(i)) for i in range(num_images))


def model_forward(model, model_name, patches, pos, scales):
    """train.py:270-275."""
    if "vtamiq" in model_name.lower():
        return model(patches, pos, scales)
    raise ValueError(f"Unsupported model {model_name}")


def predict(model, pref_module, data, is_pairwise, output_feats, use_scales):
    """train.py:278-314.  Returns (q, q_p, feats)."""
    q, patches, pos, scales = data[:4]
    if is_pairwise:
        pref, pdist1, pdist2 = split_per_image(patches)
        posref, posdist1, posdist2 = split_per_image(pos)
        scalesref, scalesdist1, scalesdist2 = split_per_image(scales) if use_scales else (None, None, None)
        if hasattr(model, "forward_pairwise"):
            # same scores as the two calls of train.py:286-287, with the shared reference image encoded once
            q1, q2 = model.forward_pairwise((pref, pdist1, pdist2), (posref, posdist1, posdist2),
                                            (scalesref, scalesdist1, scalesdist2) if use_scales else None)
            feats = (None, None) if output_feats else None
        else:
            out1 = model((pref, pdist1), (posref, posdist1), (scalesref, scalesdist1))
            out2 = model((pref, pdist2), (posref, posdist2), (scalesref, scalesdist2))
            q1, q2 = out1[0], out2[0]
            feats = (out1[1], out2[1]) if output_feats else None
        if pref_module is not None:
            q_p = pref_module(q1, q2)
        else:
            q_p = torch.sigmoid(q1 - q2)          # sign convention of train.py:301 (differs from :298 -- reproduced)
    else:
        patches = split_per_image(patches)
        pos = split_per_image(pos)
        scales = split_per_image(scales) if use_scales else (None, None)
        out = model(patches, pos, scales)
        q_p, feats = out if output_feats else (out[0], None)
    q_p = q_p.flatten()
    return q, q_p, feats

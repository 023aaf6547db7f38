"""ORACLE -- test infrastructure only (see oracle/vtamiq_oracle.py).  CPU restatement of the reference's image -> patch-tensor
step for GIVEN sample coordinates (the sampler's RNG is out of scope):

  transform_img          data/utils.py:50-96      to_tensor (uint8 HWC -> f32 CHW / 255), hflip, vflip, normalize (x - mean) / std
  get_iqa_patches        data/patch_sampling.py:529-613  per scale: pos = clamp((sample + P/2) / (dim - P/2), 0, 1-1e-6),
                                                   patch[c, i, j] = tensor[c, row + i, col + j], then AvgPool2d(2) of the image

Parity status: the gather / position / pyramid part is PINNED by tests/golden/patches_gather.npz (outputs of the reference's own
get_iqa_patches with a recording sampler).  transform_img calls torchvision (absent from the build container): its restatement
follows torchvision's documented semantics and is NOT pinned by a reference run ("parity unpinned" for that function only).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def transform_img(img_u8: np.ndarray, h_flip=False, v_flip=False, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """data/utils.py:76-94 for a uint8 HxWx3 array (torchvision.to_tensor: permute, float32, div(255); hflip flips W, vflip flips H)."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    if h_flip:
        t = t.flip(-1)
    if v_flip:
        t = t.flip(-2)
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (t - m) / s


def extract_patches(tensors: Sequence[torch.Tensor], samples_per_scale: List[Sequence[np.ndarray]], patch_dim: int = 16
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """data/patch_sampling.py:529-611.  tensors: K images (3,H,W); samples_per_scale[s][k] = int array (2, n_s) of (row, col) at
    scale s for image k (aligned sampling passes the same array for every k).  Patches are ordered by scale, scale 0 first."""
    K = len(tensors)
    num_scales = len(samples_per_scale)
    use_scales = num_scales > 1
    cur = torch.stack(list(tensors), dim=0)
    pat, pos, scl = [[] for _ in range(K)], [[] for _ in range(K)], [[] for _ in range(K)]
    half = np.array([patch_dim // 2, patch_dim // 2], np.float32).reshape(1, 2)
    ii = torch.arange(patch_dim)
    for s in range(num_scales):
        h, w = cur.shape[2:4]
        ratio = np.array([h - patch_dim // 2, w - patch_dim // 2], np.float32).reshape(1, 2)
        for k in range(K):
            smp = np.asarray(samples_per_scale[s][k])
            p = torch.from_numpy(smp).permute(1, 0)                                  # (n, 2) int
            p = torch.clamp((p + half) / ratio, 0., 1. - 1e-6)                       # :565-568
            rows = torch.from_numpy(smp[0]).view(-1, 1, 1) + ii.view(1, -1, 1)
            cols = torch.from_numpy(smp[1]).view(-1, 1, 1) + ii.view(1, 1, -1)
            pat[k].append(cur[k][:, rows, cols].permute(1, 0, 2, 3))                 # (n, 3, P, P), :539-545
            pos[k].append(p.to(torch.float32))
            scl[k].append(torch.full((smp.shape[1],), s, dtype=torch.int32))
        cur = torch.nn.functional.avg_pool2d(cur, 2)                                 # :552, 600
    patches = torch.stack([torch.cat(x) for x in pat])
    positions = torch.stack([torch.cat(x) for x in pos])
    scales = torch.stack([torch.cat(x) for x in scl]) if use_scales else None
    return patches, positions, scales

"""Seeded synthetic weights and inputs.

There is no network for checkpoints or datasets, so every test, the golden-capture script and
bench.py draw weights and patch tensors from this generator.  Every tensor gets its own legacy
`numpy.random.RandomState` stream keyed by (seed, crc32(name)), so any subset can be regenerated
independently and the stream is frozen across numpy versions.

Input conventions follow the loader contract of the reference:
  patches in [-1, 1]  (data/patch_datasets.py:51-52), pos in [0, 1-1e-6] (data/patch_sampling.py:568),
  scales = f32-cast integer scale ids ordered scale 0 first (data/patch_sampling.py:427-447, train.py:254-255).
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional, Tuple

import numpy as np

from .spec import ModelSpec


def _rs(seed: int, name: str) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def make_tensor(name: str, shape, kind: str, seed: int) -> np.ndarray:
    r = _rs(seed, name)
    if kind in ("matrix", "embed"):
        a = r.normal(0.0, 0.02, size=shape)
    elif kind == "bias":
        a = r.normal(0.0, 0.02, size=shape)
    elif kind == "ln_w":
        a = 1.0 + 0.1 * r.normal(size=shape)
    elif kind == "ln_b":
        a = 0.02 * r.normal(size=shape)
    elif kind == "gamma":
        a = 1.0 + 0.1 * r.normal(size=shape)
    elif kind == "prelu":
        a = 0.25 + 0.05 * r.normal(size=shape)
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


def make_state_dict(spec: ModelSpec, seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 numpy state dict with the reference's key names."""
    return {k: make_tensor(k, shape, kind, seed) for k, shape, kind in spec.state_layout()}


def num_patches_per_scale(patch_count: int, num_scales: int, ratio: float = 1.75) -> np.ndarray:
    """Per-scale patch counts, scale 0 first; same arithmetic as data/patch_sampling.py:427-447."""
    n = 2.0 ** (ratio * np.arange(num_scales))
    n = np.ceil(n * patch_count / np.sum(n)).astype(int)
    cum = np.cumsum(n)
    for i in range(num_scales):
        if patch_count <= cum[i]:
            n[i] -= cum[i] - patch_count
            n[i + 1:] = 0
            break
    # scale s consumes num_patches[-s-1] (data/patch_sampling.py:553): scale 0 (finest) gets the largest count
    return n[::-1].copy()


def make_inputs(spec: ModelSpec, B: int, N: int, seed: int = 1234, aligned: bool = True
                ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """Collated-batch layout of the reference loader: patches[B,2,N,3,P,P], pos[B,2,N,2], scales[B,2,N] or None.

    patches_ref ~ U(-1,1); patches_dist = clamp(ref + 0.1 N(0,1)); pos ~ U(0,1) clamped to <= 1-1e-6,
    shared by ref and dist when `aligned` (use_aligned_patches default, train_config.py:328).
    """
    P = spec.patch_size
    r = _rs(seed, f"inputs/{B}/{N}")
    if getattr(spec, "use_patch_embedding", True):
        ref = r.uniform(-1.0, 1.0, size=(B, N, 3, P, P)).astype(np.float32)
        dist = np.clip(ref + 0.1 * r.normal(size=ref.shape).astype(np.float32), -1.0, 1.0).astype(np.float32)
    else:           # a model without the patch convolution is fed pre-embedded rows: patches[B,2,N,H] (transformer.py:534-535)
        ref = (0.3 * r.normal(size=(B, N, spec.hidden_size))).astype(np.float32)
        dist = (ref + 0.03 * r.normal(size=ref.shape)).astype(np.float32)
    pos_ref = np.minimum(r.uniform(0.0, 1.0, size=(B, N, 2)), 1.0 - 1e-6).astype(np.float32)
    pos_ref = np.minimum(pos_ref, np.float32(1.0 - 1e-6))
    if aligned:
        pos_dist = pos_ref.copy()
    else:
        pos_dist = np.minimum(r.uniform(0.0, 1.0, size=(B, N, 2)), 1.0 - 1e-6).astype(np.float32)
        pos_dist = np.minimum(pos_dist, np.float32(1.0 - 1e-6))
    patches = np.stack([ref, dist], axis=1)
    pos = np.stack([pos_ref, pos_dist], axis=1)
    scales = None
    if spec.use_scale_embedding:
        counts = num_patches_per_scale(N, spec.num_scales)
        ids = np.concatenate([np.full(c, i, dtype=np.int32) for i, c in enumerate(counts)])[:N]
        scales = np.broadcast_to(ids[None, None, :], (B, 2, N)).astype(np.int32).copy()
    return patches, pos, scales


LADDER_SIGMAS = (0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5)


def make_ladder_inputs(spec: ModelSpec, images: int, N: int, seed: int = 777, sigmas=LADDER_SIGMAS
                       ) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """A distortion ladder as an IQA test set has one: `images` reference patch sets, each paired with len(sigmas) distorted
    versions dist = clamp(ref + sigma N(0,1)) -> images * len(sigmas) pairs in the collated layout of make_inputs (pair
    i*len(sigmas) + j = image i at sigmas[j]); positions shared by ref and dist (aligned sampling).  Host-side twin of bench.py's
    on-device ladder, seeded through numpy so that the golden capture (build container) and the GPU tests see the same tensors."""
    P = spec.patch_size
    S = len(sigmas)
    r = _rs(seed, f"ladder/{images}/{N}")
    ref = r.uniform(-1.0, 1.0, size=(images, N, 3, P, P)).astype(np.float32)
    pos = np.minimum(r.uniform(0.0, 1.0, size=(images, N, 2)).astype(np.float32), np.float32(1.0 - 1e-6))
    B = images * S
    patches = np.empty((B, 2, N, 3, P, P), dtype=np.float32)
    posb = np.empty((B, 2, N, 2), dtype=np.float32)
    for i in range(images):
        for j, sg in enumerate(sigmas):
            noise = r.normal(size=ref[i].shape).astype(np.float32)
            patches[i * S + j, 0] = ref[i]
            patches[i * S + j, 1] = np.clip(ref[i] + np.float32(sg) * noise, -1.0, 1.0)
            posb[i * S + j, :] = pos[i]
    scales = None
    if spec.use_scale_embedding:
        counts = num_patches_per_scale(N, spec.num_scales)
        ids = np.concatenate([np.full(c, i, dtype=np.int32) for i, c in enumerate(counts)])[:N]
        scales = np.broadcast_to(ids[None, None, :], (B, 2, N)).astype(np.int32).copy()
    return patches, posb, scales

"""the patch-extraction oracle against the golden captured from the reference's get_iqa_patches."""
import os

import numpy as np
import torch

from oracle import patch_oracle as PO
from tests.helpers import GOLDEN


PATCH_GOLDENS = {16: "patches_gather.npz", 8: "patches_gather_p8.npz"}     # patch size -> fixture (ViT-B16 / L16, ViT-B8)


def load_patch_golden(tag, P=16):
    g = dict(np.load(os.path.join(GOLDEN, PATCH_GOLDENS[P])))
    imgs = [g["img0"], g["img1"]]
    flips = tuple(bool(v) for v in g[f"{tag}/flips"])
    ncalls = int(g[f"{tag}/ncalls"])
    calls = [(tuple(g[f"{tag}/call{i}/hw"]), g[f"{tag}/call{i}/samples"]) for i in range(ncalls)]
    per_img = 1 if tag == "aligned" else 2
    nscales = ncalls // per_img
    samples = [[calls[s * per_img + (k if per_img == 2 else 0)][1] for k in range(2)] for s in range(nscales)]
    dims = [calls[s * per_img][0] for s in range(nscales)]
    return g, imgs, flips, samples, dims


import pytest


@pytest.mark.parametrize("P", [16, 8])
def test_oracle_matches_reference_get_iqa_patches(P):
    for tag in ("aligned", "unaligned"):
        g, imgs, flips, samples, dims = load_patch_golden(tag, P)
        tens = [PO.transform_img(im, flips[0], flips[1]) for im in imgs]
        patches, pos, scales = PO.extract_patches(tens, samples, patch_dim=P)
        assert patches.shape[-2:] == (P, P)
        np.testing.assert_array_equal(patches.numpy(), g[f"{tag}/patches"])
        np.testing.assert_array_equal(pos.numpy(), g[f"{tag}/pos"])
        np.testing.assert_array_equal(scales.numpy(), g[f"{tag}/scales"])
        assert dims[0] == (160, 208) and dims[1] == (80, 104) and dims[2] == (40, 52)      # AvgPool2d(2) pyramid


def test_transform_img_semantics():
    """to_tensor + flips + normalize on a hand-checkable image (torchvision semantics; not pinned by a reference run)."""
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3) * 10
    t = PO.transform_img(img)
    assert t.shape == (3, 2, 3) and t.dtype == torch.float32
    assert t[1, 0, 2].item() == (np.float32(img[0, 2, 1]) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    assert torch.equal(PO.transform_img(img, h_flip=True), t.flip(-1)) and torch.equal(PO.transform_img(img, v_flip=True), t.flip(-2))

"""JAX .npz / .pth weight ingestion against what the reference's own loaders produce (golden fixture
tests/golden/npz_ingest_tiny.npz: inputs + VisionTransformer.load_from result, incl. the 3x3 -> 4x4 position-grid resize)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN
from vtamiq_amd import weights as W


def test_convert_vit_npz_matches_reference_load_from():
    g = dict(np.load(os.path.join(GOLDEN, "npz_ingest_tiny.npz")))
    H, L, ntok = (int(v) for v in g["meta"])
    src = {k[3:]: v for k, v in g.items() if k.startswith("in/")}
    want = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    got = W.convert_vit_npz(src, H, L, ntok)
    untouched = {"transformer.embeddings.extra_tokens"}            # not part of a ViT checkpoint (transformer.py:650-654)
    assert set(got) == set(want) - untouched
    for k, v in got.items():
        assert v.dtype == torch.float32 and tuple(v.shape) == want[k].shape, k
        np.testing.assert_allclose(v.numpy(), want[k], rtol=0, atol=1e-6, err_msg=k)     # bit-exact but for the zoom's float path


def test_resize_is_identity_when_sizes_match():
    p = np.random.RandomState(0).randn(1, 17, 8).astype(np.float32)
    assert W.resize_pos_embedding(p, 17) is p


def test_load_vit_npz_and_checkpoint_into_model(tmp_path):
    """Round trip on the real ViT-B/16 layout (truncated to 1 layer for speed): npz -> model -> .pth -> model."""
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1), num_rgs=1, num_rcabs=1, precision="bf16")
    rs = np.random.RandomState(1)
    H, M, P = 768, 3072, 16
    w = {"embedding/kernel": rs.randn(P, P, 3, H), "embedding/bias": rs.randn(H), "cls": rs.randn(1, 1, H),
         "Transformer/posembed_input/pos_embedding": rs.randn(1, 577, H), "Transformer/encoder_norm/scale": rs.randn(H),
         "Transformer/encoder_norm/bias": rs.randn(H)}
    r = "Transformer/encoderblock_0"
    for nm in ("query", "key", "value"):
        w[f"{r}/MultiHeadDotProductAttention_1/{nm}/kernel"] = rs.randn(H, 12, 64)
        w[f"{r}/MultiHeadDotProductAttention_1/{nm}/bias"] = rs.randn(12, 64)
    w[f"{r}/MultiHeadDotProductAttention_1/out/kernel"] = rs.randn(12, 64, H)
    w[f"{r}/MultiHeadDotProductAttention_1/out/bias"] = rs.randn(H)
    w[f"{r}/MlpBlock_3/Dense_0/kernel"] = rs.randn(H, M); w[f"{r}/MlpBlock_3/Dense_0/bias"] = rs.randn(M)
    w[f"{r}/MlpBlock_3/Dense_1/kernel"] = rs.randn(M, H); w[f"{r}/MlpBlock_3/Dense_1/bias"] = rs.randn(H)
    for ln in ("LayerNorm_0", "LayerNorm_2"):
        w[f"{r}/{ln}/scale"] = rs.randn(H); w[f"{r}/{ln}/bias"] = rs.randn(H)
    w = {k: v.astype(np.float32) for k, v in w.items()}
    path = tmp_path / "vit.npz"
    np.savez(path, **w)
    head_before = m.q_predictor[1].weight.clone()
    W.load_vit_npz(m, str(path))
    sd = m.state_dict()
    # torch Linear layout: y = x W^T  <=>  W = kernel.reshape(H, H).T
    np.testing.assert_array_equal(sd["transformer.encoder.layers.0.attn.query.weight"].numpy(),
                                  w[f"{r}/MultiHeadDotProductAttention_1/query/kernel"].reshape(H, H).T)
    np.testing.assert_array_equal(sd["transformer.embeddings.patch_embeddings.weight"].numpy(),
                                  w["embedding/kernel"].transpose(3, 2, 0, 1))
    assert torch.equal(m.q_predictor[1].weight, head_before)              # the head is not in a ViT checkpoint
    # .pth in the reference's format, incl. dropping the transformer keys and the non-strict fallback
    ck = tmp_path / "best.pth"
    torch.save({"epoch": 3, "SROCC": 0.9, W.MODEL_STATE_DICT: m.state_dict()}, ck)
    m2 = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1), num_rgs=1, num_rcabs=1, precision="bf16")
    out = W.load_checkpoint(m2, str(ck))
    assert out["epoch"] == 3
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    m3 = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1), num_rgs=1, num_rcabs=1, precision="bf16")
    before = m3.transformer.encoder.layers[0].attn.query.weight.clone()
    with pytest.warns(UserWarning, match="partial load"):
        W.load_checkpoint(m3, str(ck), allow_vit=False)
    assert torch.equal(m3.transformer.encoder.layers[0].attn.query.weight, before)       # transformer.* dropped
    assert torch.equal(m3.q_predictor[1].weight, m.q_predictor[1].weight)                # head loaded

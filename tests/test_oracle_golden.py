"""Pin the oracle (CPU restatement) against golden vectors captured from the imported reference."""
import numpy as np
import pytest
import torch

from oracle import vtamiq_oracle as O
from tests.helpers import E2E_CASES, FULLSIZE_CASES, LADDER_CASES, LONG_CASES, OPERATING_POINT_CASES, STRESS_CASES, GOLDEN, gate_error, load_case, load_ladder_case, split_inputs, rel_err
import os

# fp32 op-order differences between the restatement and the reference modules stay below this
ORACLE_RTOL = 2e-5


@pytest.mark.parametrize("name", E2E_CASES)
def test_e2e_q(name):
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales)
    q, aux = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)
    assert aux is None and q.shape == (int(g["B"]),) and q.dtype == torch.float32
    e = rel_err(q.numpy(), g["q"])
    assert e["max_rel_rms"] < ORACLE_RTOL, e
    assert e["max_abs"] < 1e-6, e


def test_e2e_q_from_a_register_token():
    """VTAMIQ.token_num (vtamiq.py:57, 107-108): the reference with a register token as the IQA token."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("token2_b3_n45")
    p, ps, sc = split_inputs(patches, pos, scales)
    q, _ = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc, token_num=int(g["token_num"]))
    e = rel_err(q.numpy(), g["q_token"])
    assert e["max_rel_rms"] < ORACLE_RTOL and e["max_abs"] < 1e-6, e
    assert np.abs(g["q_token"] - g["q"]).max() > 1e-3        # a different token gives different scores


@pytest.mark.parametrize("name", FULLSIZE_CASES)
def test_e2e_q_at_the_bench_sizes(name):
    """The reference's scores at the sizes bench.py runs (B = 32, N = 500, L = 12; reference-default topology B = 16, N = 512) and at BASELINE
    configs[3] (ViT-L/16, B = 16, N = 1024, 3 scales): a bounded
    sample here (pairs are independent: the first 3), the GPU test scores the whole batch.  Bound: on the flat init the B = 32, L = 12
    scores are small through cancellation (rms 7e-3) and BOTH fp32 evaluations sit 4e-5 of that rms from the float64 scores (reference
    3.7e-5, oracle 4.6e-5, measured over 4 pairs), so fp32 against fp32 is gated at 1.5e-4 of the rms; the L = 6 case meets 2e-5."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    n = 1 if name.startswith("c4_") else 3          # one ViT-L pair at N = 1024 is 1.4 TFLOP of host work
    p, ps, sc = split_inputs(patches[:n], pos[:n], scales[:n] if scales is not None else None)
    q = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)[0].numpy()
    e = rel_err(q, g["q"][:n])
    tol = 1.5e-4 if name == "c2_b32_n500" else ORACLE_RTOL
    assert e["max_abs"] < tol * float(np.sqrt(np.mean(g["q"] ** 2))) and e["max_abs"] < 1e-6, e


@pytest.mark.parametrize("name", LONG_CASES)
def test_e2e_q_long_sequence(name):
    """The long-sequence regime (reference README.md:85: 5000 patches; S = 5001, attention > 50 % of the flops): the oracle against the
    reference's scores at ViT-B/16 L = 12, N = 5000 -- one pair here (~20 s of host work), the GPU test scores every pair.  The trained-like
    case goes through a head at its operating point (score 0.45) and carries the reference's float64 score too: raw relative error."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches[:1], pos[:1], None)
    q = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)[0].numpy()
    e = rel_err(q, g["q"][:1])
    print(name, "oracle32-ref32", e["max_rel"], "" if "q64" not in g else ("oracle32-ref64", rel_err(q, g["q64"][:1])["max_rel"], "ref32-ref64", rel_err(g["q"][:1], g["q64"][:1])["max_rel"]))
    assert e["max_rel"] < 1e-4, e
    if "q64" in g:
        assert rel_err(q, g["q64"][:1])["max_rel"] < 1e-4 and rel_err(g["q"], g["q64"])["max_rel"] < 1e-4


@pytest.mark.parametrize("name", STRESS_CASES)
def test_e2e_q_trained_like_statistics(name):
    """The oracle against the REFERENCE run on stress_state weights (peaked softmax, outlier channels).
    Evaluated in float64 the restatement reproduces the reference's float64 scores to 1e-12 (it IS the same algorithm).  In float32
    two evaluations of this model differ through their operation order alone: the reference is 1.2e-5 (qk = 3) / 9.0e-5 (qk = 5)
    from its own float64 scores, the oracle 8.8e-6 / 4.9e-5; the fp32-vs-fp32 bound is therefore 3e-4 raw, and each against
    float64 1.5e-4."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales)
    q, aux = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)
    e = rel_err(q.numpy(), g["q"])
    e64 = rel_err(q.numpy(), g["q64"])
    r64 = rel_err(g["q"], g["q64"])
    print(name, "oracle32-ref32", e["max_rel"], "oracle32-ref64", e64["max_rel"], "ref32-ref64", r64["max_rel"])
    assert e["max_rel"] < 3e-4 and e64["max_rel"] < 1.5e-4 and r64["max_rel"] < 1.5e-4, (e, e64, r64)
    sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    p64, ps64, sc64 = split_inputs(patches, pos, scales, dtype=torch.float64)
    q64 = O.vtamiq_forward(sd64, spec, p64, ps64, sc64)[0]
    assert q64.dtype == torch.float64
    assert rel_err(q64.numpy(), g["q64"])["max_rel"] < 1e-10


@pytest.mark.parametrize("name", LADDER_CASES)
def test_ladder_scores_at_baseline_patch_count(name):
    """The oracle against the reference's fp32 and float64 scores of the 64-pair N = 500 ladder on trained-like weights (a bounded
    sample here: the first 8 pairs in fp32 and in float64; the GPU test scores all 64).  float64: the same algorithm, 1e-10."""
    g, kw, spec, sd, (patches, pos, scales) = load_ladder_case(name)
    n = 8
    p, ps, sc = split_inputs(patches[:n], pos[:n], None)
    q = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)[0].numpy()
    # fp32 against fp32 / fp64 of the reference in the gate's measure (raw relative for |q| >= 0.1 rms of the sample)
    e32, e64 = gate_error(q, g["q"][:n]), gate_error(q, g["q64"][:n])
    print(name, "oracle32-ref32", e32, "oracle32-ref64", e64)
    assert e32 < 5e-4 and e64 < 5e-4, (e32, e64)
    sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    p64, ps64, _ = split_inputs(patches[:n], pos[:n], None, dtype=torch.float64)
    q64 = O.vtamiq_forward(sd64, spec, p64, ps64, (None, None))[0].numpy()
    assert np.max(np.abs(q64 - g["q64"][:n])) < 1e-10 * np.sqrt(np.mean(g["q64"] ** 2)) + 1e-12


@pytest.mark.parametrize("name", OPERATING_POINT_CASES)
def test_operating_point_ladder(name):
    """The oracle against the reference on the ladder scored through a head at a trained model's operating point (scores in [0.2, 0.8],
    tests.helpers.stress_state(head=True)): RAW relative error of EVERY score of a bounded sample (first 8 pairs), no rms floor; float64
    is the same algorithm to 1e-12."""
    g, kw, spec, sd, (patches, pos, scales) = load_ladder_case(name)
    assert int(g["stress_head"]) == 1 and 0.2 < float(g["q64"].min()) and float(g["q64"].max()) < 0.8 and float(g["q64"].std()) > 0.05
    n = 8
    p, ps, sc = split_inputs(patches[:n], pos[:n], None)
    q = O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc)[0].numpy()
    e32, e64 = rel_err(q, g["q"][:n])["max_rel"], rel_err(q, g["q64"][:n])["max_rel"]
    print(name, "oracle32-ref32", e32, "oracle32-ref64", e64)
    assert e32 < 1e-4 and e64 < 1e-4, (e32, e64)
    sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    p64, ps64, _ = split_inputs(patches[:n], pos[:n], None, dtype=torch.float64)
    q64 = O.vtamiq_forward(sd64, spec, p64, ps64, (None, None))[0].numpy()
    assert np.max(np.abs(q64 - g["q64"][:n])) < 1e-12


def test_per_layer_tokens_c1():
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales)
    trace = {}
    O.vtamiq_forward(O.to_torch(sd), spec, p, ps, sc, trace=trace)
    for side in ("ref", "dist"):
        got = trace[f"tokens_{side}"][1:].numpy()          # entry 0 is the embedding output
        want = g[f"tokens_{side}"]                         # (L,B,T,H) pre final LN, return_layers=True
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5)


def _sub(ops, prefix):
    return {k[len(prefix) + 4:]: torch.from_numpy(v) for k, v in ops.items() if k.startswith(prefix + "/sd/")}


def test_toy_ops():
    ops = dict(np.load(os.path.join(GOLDEN, "ops_toy.npz")))
    T = lambda k: torch.from_numpy(ops[k])
    # attention (transformer.py:153-172) incl. the returned probabilities
    sd = {"L.attn." + k: v for k, v in _sub(ops, "mhsa").items()}
    y, probs = O.attention(sd, "L.", T("mhsa/x"), 4, return_probs=True)
    np.testing.assert_allclose(y.numpy(), ops["mhsa/y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(probs.numpy(), ops["mhsa/probs"], rtol=1e-5, atol=1e-7)
    # MLP with exact-erf GELU
    sd = {"L.ffn." + k: v for k, v in _sub(ops, "mlp").items()}
    np.testing.assert_allclose(O.mlp(sd, "L.", T("mlp/x")).numpy(), ops["mlp/y"], rtol=1e-5, atol=1e-6)

    # full encoder layer with LayerScale
    class S:  # minimal spec view
        num_heads, use_layer_scale = 4, True
    sd = {"transformer.encoder.layers.0." + k: v for k, v in _sub(ops, "layer").items()}
    np.testing.assert_allclose(O.encoder_layer(sd, S, 0, T("layer/x")).numpy(), ops["layer/y"], rtol=1e-5, atol=1e-6)

    # embeddings: patch conv + pos gather + scale gather (clamped) + CLS(+pos[0]) + 2 register tokens
    class E:
        pos_grid, num_scales, use_scale_embedding, num_extra_tokens = 4, 3, True, 2
    sd = {"transformer.embeddings." + k: v for k, v in _sub(ops, "emb").items()}
    y = O.embeddings(sd, E, T("emb/patches"), T("emb/pos"), T("emb/scales"))
    np.testing.assert_allclose(y.numpy(), ops["emb/y"], rtol=1e-5, atol=5e-6)   # conv vs GEMM summation order, K=768
    # uv position index edge cases (0, 1-1e-6, exact cell borders)
    table = torch.from_numpy(ops["uvpos/sd/positional_embeddings"])[0]
    got = table[O.pos_index(T("uvpos/pos"), 24)]
    np.testing.assert_array_equal(got.numpy(), ops["uvpos/y"][0])
    # RCAB / ResidualGroup on (B,C,1)
    sd = {"R." + k: v for k, v in _sub(ops, "rcab").items()}
    np.testing.assert_allclose(O.rcab(sd, "R.", T("rcab/x")[..., 0]).numpy(), ops["rcab/y"][..., 0], rtol=1e-5, atol=1e-6)

    class G:
        calibrate, num_rgs, num_rcabs = True, 1, 2
    sd = {"quality_decoder.0." + k: v for k, v in _sub(ops, "rg").items()}
    x = T("rg/x")[..., 0]
    y = x
    for k in range(2):
        y = O.rcab(sd, f"quality_decoder.0.body.{k}.", y)
    y = x + O._conv1x1(sd, "quality_decoder.0.body.2", y)
    np.testing.assert_allclose(y.numpy(), ops["rg/y"][..., 0], rtol=1e-5, atol=1e-6)


def test_plumbing_c1():
    """BASELINE config 1: collated batch -> f32 cast -> per-image split -> model -> (q, q_p)."""
    import json
    from vtamiq_amd import synth
    from vtamiq_amd.spec import make_spec
    g = dict(np.load(os.path.join(GOLDEN, "plumbing_c1.npz")))
    spec = make_spec(**json.loads(str(g["kwargs"])))
    sd = O.to_torch(synth.make_state_dict(spec, int(g["wseed"])))
    patches, pos, _ = synth.make_inputs(spec, int(g["B"]), int(g["N"]), int(g["iseed"]))
    batch = (g["q_in"], patches, pos, np.full((int(g["B"]),), -1, dtype=np.int32))
    q, q_p = O.predict(sd, spec, batch)
    assert q.dtype == torch.float32 and q_p.shape == (2,)
    np.testing.assert_array_equal(q.numpy(), g["q"])
    assert rel_err(q_p.numpy(), g["q_p"])["max_rel_rms"] < ORACLE_RTOL


def test_fp64_agrees():
    """fp32 oracle vs the same restatement in fp64: the fp32 noise floor that bounds any parity claim."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales, dtype=torch.float64)
    q64, _ = O.vtamiq_forward(O.to_torch(sd, torch.float64), spec, p, ps, sc)
    e = rel_err(g["q"], q64.numpy())
    assert e["max_rel_rms"] < 1e-4, e

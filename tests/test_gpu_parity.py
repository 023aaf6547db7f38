"""End-to-end parity of the HIP path (through the drop-in VTAMIQ callable and the C ABI) against
  (a) the golden vectors captured from the imported reference (tests/golden/*.npz), and
  (b) the oracle run on the host on the same seeded inputs,
plus size-independent properties at BASELINE.json's full size (B=32, N=500).

Tolerance (BASELINE.json north_star): scores within 1e-3 RELATIVE of the fp32 CPU reference.  The gate applies to
precision="bf16x3".  Per-element relative error is reported raw; because random-init scores cross zero
(|q| down to 6e-4 against an rms of ~3e-2) the gate is evaluated with the denominator max(|q_ref|, rms(q_ref)).
precision="bf16" (single-MFMA throughput mode) is measured against a looser bound that is stated here, not hidden:
1e-1 (same denominator) -- SURVEY.md section 7 measured 4.5e-2 max relative for bf16 operand rounding on this model with
random-init weights, and the golden cases here land between 7e-3 and 6e-2.  It is a smoke bound against gross errors,
NOT a parity claim: only bf16x3 claims the north-star tolerance.
"""
import json

import numpy as np
import pytest
import torch

from oracle import vtamiq_oracle as O
from tests.helpers import E2E_CASES, load_case, rel_err, split_inputs, stress_state
from vtamiq_amd import VTAMIQ, synth
from vtamiq_amd.predict import get_data_tuple, predict

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"bf16x3": 1e-3, "bf16": 1e-1}


def gate(q, q_ref, tol):
    q, q_ref = np.asarray(q, np.float64), np.asarray(q_ref, np.float64)
    rms = np.sqrt(np.mean(q_ref ** 2))
    return float(np.max(np.abs(q - q_ref) / np.maximum(np.abs(q_ref), rms))) < tol


def build(kw, sd_np, precision):
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("name", E2E_CASES)
def test_golden(name, precision):
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd, precision)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q, aux = model(p, ps, sc)
    assert aux is None and q.shape == (int(g["B"]),) and q.dtype == torch.float32 and q.device.type == "cuda"
    e = rel_err(q.cpu().numpy(), g["q"])
    print(f"\n[{name} {precision}] {e}")
    assert np.isfinite(q.cpu().numpy()).all()
    assert gate(q.cpu().numpy(), g["q"], TOL[precision]), e


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_token_trace_c1(precision):
    """Per-layer CLS rows (pre final LN) against the reference's return_layers=True capture: localises any divergence."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    model = build(kw, sd, precision)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    B, L, T, H = int(g["B"]), spec.num_layers, spec.num_tokens, spec.hidden_size
    trace = torch.zeros(L + 1, 2 * B, T, H, device=DEV)
    with torch.no_grad():
        model(p, ps, sc, _trace=trace)
    got = trace.cpu().numpy()
    want = np.concatenate([g["tokens_ref"], g["tokens_dist"]], axis=1)       # (L, 2B, T, H)
    tol = 2e-4 if precision == "bf16x3" else 4e-2
    for layer in range(L):
        d = np.abs(got[layer + 1] - want[layer]).max() / np.abs(want[layer]).max()
        assert d < tol, (layer, d)


def test_plumbing_c1():
    """BASELINE config 1 through the train.py-equivalent boundary: collated batch -> get_data_tuple -> predict."""
    import os
    from tests.helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "plumbing_c1.npz")))
    kw = json.loads(str(g["kwargs"]))
    model = VTAMIQ(**json.loads(json.dumps(kw)), precision="bf16x3")
    spec = model.spec
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec, int(g["wseed"])).items()})
    model = model.to(DEV).eval()
    patches, pos, _ = synth.make_inputs(spec, int(g["B"]), int(g["N"]), int(g["iseed"]))
    batch = (torch.from_numpy(g["q_in"]), torch.from_numpy(patches), torch.from_numpy(pos),
             torch.full((int(g["B"]),), -1, dtype=torch.int32))
    with torch.no_grad():
        data = get_data_tuple(batch, torch.device(DEV))
        q, q_p, feats = predict(model, None, data, False, False, False)
    assert feats is None and q.dtype == torch.float32
    np.testing.assert_array_equal(q.cpu().numpy(), g["q"])
    assert gate(q_p.cpu().numpy(), g["q_p"], 1e-3), rel_err(q_p.cpu().numpy(), g["q_p"])


def _c2_model(precision):
    kw = dict(vit_config=dict(variant="ViT-B16"))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    sd = synth.make_state_dict(m.spec, 5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_full_size_properties(precision):
    """BASELINE config 2 shape (B=32, N=500, ViT-B/16, L=12)."""
    model, sd = _c2_model(precision)
    spec = model.spec
    B, N = 32, 500
    patches, pos, _ = synth.make_inputs(spec, B, N, 4321)
    p, ps, sc = split_inputs(patches, pos, None, device=DEV)
    with torch.no_grad():
        q = model(p, ps, sc)[0]
        # (1) batch invariance: a pair's score does not depend on its position in / the size of the batch (bitwise)
        sel = [0, 13, 31]
        q_small = model(tuple(t[sel].contiguous() for t in p), tuple(t[sel].contiguous() for t in ps), (None, None))[0]
        assert torch.equal(q[sel], q_small)
        # (2) determinism
        assert torch.equal(q, model(p, ps, sc)[0])
        # (3) identical ref and dist -> zero CLS difference -> every pair gets the same score, head(0)
        q_same = model((p[0], p[0]), (ps[0], ps[0]), (None, None))[0]
        assert torch.equal(q_same, q_same[:1].expand(B))
    zero = torch.zeros(1, 1, spec.hidden_size)
    q0 = O.head(O.to_torch(sd), spec, zero, zero)
    assert abs(float(q_same[0]) - float(q0[0])) < 1e-6 + 1e-5 * abs(float(q0[0]))
    # (4) oracle on the host for a few pairs of the full-size batch (exact same inputs)
    cp, cps, _ = split_inputs(patches[sel], pos[sel], None)
    q_ref = O.vtamiq_forward(O.to_torch(sd), spec, cp, cps, (None, None))[0].numpy()
    e = rel_err(q[sel].cpu().numpy(), q_ref)
    print(f"\n[full-size {precision}] {e}")
    assert gate(q[sel].cpu().numpy(), q_ref, TOL[precision]), e


@pytest.mark.parametrize("name", ["c1_b2_n50", "refdefault_b2_n64", "vitl_b2_n70"])
def test_cls_pruned_last_layer_matches_full_layer(name, monkeypatch):
    """The CLS-only tail of the last layer (cls_tail.hip) against running the full last layer (VTQ_NO_CLS_PRUNE=1)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q_pruned = build(kw, sd, "bf16x3")(p, ps, sc)[0].cpu().numpy()
        monkeypatch.setenv("VTQ_NO_CLS_PRUNE", "1")
        q_full = build(kw, sd, "bf16x3")(p, ps, sc)[0].cpu().numpy()
    assert gate(q_pruned, g["q"], 1e-3) and gate(q_full, g["q"], 1e-3)
    assert gate(q_pruned, q_full, 3e-4), rel_err(q_pruned, q_full)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_pairwise_triplets(precision):
    """(ref, dist1, dist2) items.  The fused entry point encodes ref once and must reproduce the two model calls of
    train.py:286-287 bit for bit; predict() then applies the PreferenceModule / sigmoid exactly like train.py:296-301."""
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=3, num_scales=3))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    sd = synth.make_state_dict(m.spec, 31)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    B, N = 4, 60
    patches, pos, scales = synth.make_inputs(m.spec, B, N, 32, aligned=False)
    rs = np.random.RandomState(3)
    p3 = np.concatenate([patches, np.clip(patches[:, :1] + 0.2 * rs.randn(*patches[:, :1].shape), -1, 1).astype(np.float32)], axis=1)
    pos3 = np.concatenate([pos, pos[:, 1:2]], axis=1)
    sc3 = np.concatenate([scales, scales[:, 1:2]], axis=1)
    q_true = torch.zeros(B, dtype=torch.float64)
    batch = (q_true, torch.from_numpy(p3), torch.from_numpy(pos3), torch.from_numpy(sc3))
    from vtamiq_amd import PreferenceModule
    pref = PreferenceModule(1.7).to(DEV)
    with torch.no_grad():
        data = get_data_tuple(batch, torch.device(DEV))
        q, q_p, feats = predict(m, pref, data, True, False, True)
        pr, pd1, pd2 = (data[1][:, i].clone() for i in range(3))
        qr, qd1, qd2 = (data[2][:, i].clone() for i in range(3))
        sr, sd1, sd2 = (data[3][:, i].clone() for i in range(3))
        q1 = m((pr, pd1), (qr, qd1), (sr, sd1))[0]
        q2 = m((pr, pd2), (qr, qd2), (sr, sd2))[0]
        f1, f2 = m.forward_pairwise((pr, pd1, pd2), (qr, qd1, qd2), (sr, sd1, sd2))
    assert torch.equal(f1, q1) and torch.equal(f2, q2)
    assert torch.equal(q_p, torch.sigmoid(1.7 * (q2 - q1)))
    # and against the oracle's two forward calls
    c = lambda t: t.cpu()
    o1 = O.vtamiq_forward(O.to_torch(sd), m.spec, (c(pr), c(pd1)), (c(qr), c(qd1)), (c(sr), c(sd1)))[0].numpy()
    o2 = O.vtamiq_forward(O.to_torch(sd), m.spec, (c(pr), c(pd2)), (c(qr), c(qd2)), (c(sr), c(sd2)))[0].numpy()
    assert gate(np.concatenate([f1.cpu().numpy(), f2.cpu().numpy()]), np.concatenate([o1, o2]), TOL[precision])


def test_rejects_what_the_reference_rejects():
    model, _ = _c2_model("bf16")
    p = torch.zeros(1, 8, 3, 16, 16, device=DEV)
    pos = torch.zeros(1, 8, 2, device=DEV)
    model.train()
    with pytest.raises(NotImplementedError):
        model((p, p), (pos, pos), (None, None))
    model.eval()
    with pytest.raises(RuntimeError):
        model((p.cpu(), p.cpu()), (pos.cpu(), pos.cpu()), (None, None))
    ms = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, num_scales=3), precision="bf16").to(DEV).eval()
    with pytest.raises(ValueError, match="scales is passed as None"):
        ms((p, p), (pos, pos), (None, None))


def test_weight_reload_is_seen():
    """load_state_dict after the first forward must reach the engine (drop-in semantics)."""
    model, sd = _c2_model("bf16")
    spec = model.spec
    patches, pos, _ = synth.make_inputs(spec, 2, 20, 5)
    p, ps, sc = split_inputs(patches, pos, None, device=DEV)
    with torch.no_grad():
        q1 = model(p, ps, sc)[0].clone()
        sd2 = synth.make_state_dict(spec, 6)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
        q2 = model(p, ps, sc)[0].clone()
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        q3 = model(p, ps, sc)[0]
    assert not torch.equal(q1, q2)
    assert torch.equal(q1, q3)


@pytest.mark.parametrize("B,N,extra,variant,scales", [(1, 8, 0, "ViT-B16", 0), (5, 77, 0, "ViT-B16", 0), (3, 200, 8, "ViT-B16", 0),
                                                      (1, 130, 0, "ViT-L16", 3), (7, 56, 3, "ViT-B16", 2), (2, 119, 8, "ViT-B16", 0)])
@pytest.mark.parametrize("parts", ["1", "2"])
def test_ragged_shapes_against_oracle(B, N, extra, variant, scales, parts, monkeypatch):
    """Edge shapes: B = 1 / odd B (part-batches fall back to one stream), S = N + T hitting 9 / 78 / 209 / 128 exactly (S == S_pad),
    register tokens, 2- and 3-scale embeddings, ViT-L; oracle on the host as the checker."""
    monkeypatch.setenv("VTQ_PARTS", parts)
    kw = dict(vit_config=dict(variant=variant, num_keep_layers=2, num_extra_tokens=extra, num_scales=scales, use_layer_scale=bool(extra)),
              num_rgs=2, num_rcabs=2, ca_reduction=16)
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="bf16x3")
    sd = synth.make_state_dict(m.spec, 40 + B)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    patches, pos, sc = synth.make_inputs(m.spec, B, N, 50 + N, aligned=bool(B & 1))
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    with torch.no_grad():
        q = m(p, ps, s3)[0].cpu().numpy()
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(O.to_torch(sd), m.spec, cp, cps, cs)[0].numpy()
    assert q.shape == (B,) and np.isfinite(q).all()
    assert gate(q, q_ref, 1e-3), rel_err(q, q_ref)


@pytest.mark.parametrize("qk", [3.0, 5.0, 8.0])
def test_trained_like_statistics_against_oracle(qk):
    """The flat random init makes attention uniform and activations small, which flatters reduced-precision operands.  Same
    topology (ViT-B/16, all 12 layers) with peaked softmax rows (mean max-probability ~0.6 / ~0.85 / ~0.95) and outlier channels
    of ~30x the stream's rms.  Near one-hot attention makes the MODEL ill-conditioned: the oracle itself moves by `cond` between
    fp32 and fp64 arithmetic (5e-6 / 2.5e-4 / 5e-4 here), i.e. it amplifies a 6e-8 rounding by up to 1e4.  bf16x3 operands
    carry 2^-17, so the bound that can be asked of it is max(1e-3, 30 * cond): the north-star 1e-3 wherever the fp32 reference
    is itself reproducible to ~3e-5, proportionally more where it is not.  The single-MFMA mode's error is printed."""
    kw = dict(vit_config=dict(variant="ViT-B16"))
    spec = VTAMIQ(**json.loads(json.dumps(kw))).spec
    sd = stress_state(spec, 5, qk=qk)
    patches, pos, sc = synth.make_inputs(spec, 3, 90, 9)
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(O.to_torch(sd), spec, cp, cps, cs)[0].numpy()
    q_f64 = O.vtamiq_forward(O.to_torch(sd, dtype=torch.float64), spec, [t.double() for t in cp], [t.double() for t in cps], cs)[0].numpy()
    cond = rel_err(q_ref, q_f64)["max_rel_rms"]
    tol = max(1e-3, 30.0 * cond)
    errs = {}
    for precision in ("bf16x3", "bf16"):
        m = build(kw, sd, precision)
        with torch.no_grad():
            q = m(p, ps, s3)[0].cpu().numpy()
        assert np.isfinite(q).all()
        errs[precision] = rel_err(q, q_ref)["max_rel_rms"]
        if precision == "bf16x3":
            assert gate(q, q_ref, tol), (q, q_ref, errs, cond)
    print("trained-like statistics qk=%g: oracle fp32-vs-fp64 %.1e, tolerance %.1e, errors %s" % (qk, cond, tol, errs))

"""End-to-end parity of the HIP path (through the drop-in VTAMIQ callable and the C ABI) against
  (a) the golden vectors captured from the imported reference (tests/golden/*.npz), and
  (b) the oracle run on the host on the same seeded inputs,
plus size-independent properties at BASELINE.json's full size (B=32, N=500).

Tolerance (BASELINE.json north_star): scores within 1e-3 RELATIVE of the fp32 CPU reference.  The gate is the RAW per-score
relative error |q - q_ref| / |q_ref| wherever |q_ref| >= 0.1 rms(q_ref); random-init scores cross zero (one golden score is 6e-4
against an rms of 2e-2), and for those near-zero scores the denominator is rms(q_ref) (helpers.gate_error).
profiles/archive/r02_golden_errors.txt holds the raw table of every case and mode (tools/golden_errors.py).
Only precision="fp16x3" (the model's default) claims the north-star tolerance: its worst golden case is 4e-5.  The other modes
are measured against looser bounds that are stated here, not hidden -- smoke bounds against gross errors, NOT parity claims:
  "bf16x3" 3e-3 (golden cases <= 3.2e-4; the round-1 parity mode, kept: it misses 1e-3 raw on small scores of ViT-L cases),
  "fp16x2" 5e-3 (weights in single fp16: golden cases <= 2e-3), "fp16" 3e-2 (<= 7.4e-3), "bf16" 2.5e-1 (golden cases <= 8.3e-2; small
  scores of other seeded cases reach 1.4e-1).
"""
import json

import numpy as np
import pytest
import torch

from oracle import vtamiq_oracle as O
from tests.helpers import E2E_CASES, FULLSIZE_CASES, LADDER_CASES, LONG_CASES, OPERATING_POINT_CASES, STRESS_CASES, gate_error, load_case, load_ladder_case, rel_err, split_inputs, stress_state
from vtamiq_amd import VTAMIQ, _lib, synth
from vtamiq_amd.predict import get_data_tuple, predict

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"fp16x3": 1e-3, "bf16x3": 3e-3, "fp16x2": 5e-3, "fp16": 3e-2, "bf16": 2.5e-1}
ALL_MODES = ["fp16x3", "bf16x3", "fp16x2", "fp16", "bf16"]
MAIN = "fp16x3"          # the default precision of the drop-in model


def gate(q, q_ref, tol):
    return gate_error(q, q_ref) < tol


def build(kw, sd_np, precision, engine_options=0):
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision, engine_options=engine_options)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("precision", ALL_MODES)
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


@pytest.mark.parametrize("options", [0, _lib.OPT_FULL_LAST_LAYER])
@pytest.mark.parametrize("precision", ["auto", "bf16x3", "fp16"])
def test_token_num_selects_the_iqa_token(precision, options):
    """model.token_num (vtamiq.py:57, 107-108: "can be CLS token or extra_token") is read at every forward, like the reference reads
    it: register token 2 gives the reference's scores for that token (pruned and full last layer), back to 0 the CLS scores, and an
    index that is not a token of the model is refused (the reference would silently read a patch row)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("token2_b3_n45")
    model = build(kw, sd, precision, engine_options=options)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    tol = TOL["fp16x3" if precision == "auto" else precision]
    with torch.no_grad():
        q0 = model(p, ps, sc)[0].cpu().numpy()
        model.token_num = int(g["token_num"])
        q2 = model(p, ps, sc)[0].cpu().numpy()
        q2p = torch.cat(model.forward_pairwise((p[0], p[1], p[1]), (ps[0], ps[1], ps[1]), None if sc[0] is None else (sc[0], sc[1], sc[1]))).cpu().numpy()
        model.token_num = 0
        q0b = model(p, ps, sc)[0].cpu().numpy()
        model.token_num = spec.num_tokens
        with pytest.raises(RuntimeError, match="vtq_set_iqa_token"):
            model(p, ps, sc)
    assert gate(q0, g["q"], tol), rel_err(q0, g["q"])
    assert gate(q2, g["q_token"], tol), rel_err(q2, g["q_token"])
    assert np.array_equal(q0, q0b) and np.array_equal(np.concatenate([q2, q2]), q2p)


def test_pre_embedded_input():
    """Embeddings.forward's (B, N, H) branch (transformer.py:527-535): a model WITH the patch convolution handed 3-D rows skips it, like the reference
    (same weights otherwise: the oracle's scores); a model WITHOUT it handed 5-D patches fails on the missing module, like the reference; the pairwise
    entry takes the same 3-D rows (vtq_forward_pairwise_tokens) and returns the bits of two plain calls; a negative `token_num` counts from the
    last token like Python indexing into the reference's (B, H, T) rows (vtamiq.py:107-108)."""
    g, kw, spec, sd, (feats, pos, scales) = load_case("preemb_b3_n60")
    kw2 = json.loads(json.dumps(kw)); kw2["vit_config"].pop("use_patch_embedding")
    spec2 = VTAMIQ(**json.loads(json.dumps(kw2)), precision=MAIN).spec
    sd2 = dict(synth.make_state_dict(spec2, 77))
    sd2.update(sd)                                           # the golden's weights + a patch convolution that must not matter
    model = build(kw2, sd2, "auto")
    p, ps, sc = split_inputs(feats, pos, scales, device=DEV)
    with torch.no_grad():
        q, _ = model(p, ps, sc)
    assert gate(q.cpu().numpy(), g["q"], TOL["fp16x3"]), rel_err(q.cpu().numpy(), g["q"])
    nopatch = build(kw, sd, MAIN)
    with torch.no_grad():
        assert torch.equal(nopatch(p, ps, sc)[0], q)        # the same engine path, the same bits
        with pytest.raises(AttributeError, match="patch_embeddings"):
            nopatch((torch.zeros(3, 60, 3, 16, 16, device=DEV),) * 2, ps, sc)
        for m in (model, nopatch):
            q1, q2 = m.forward_pairwise((p[0], p[1], p[1]), (ps[0], ps[1], ps[1]), (sc[0], sc[1], sc[1]))
            assert torch.equal(q1, q) and torch.equal(q2, q)
        with pytest.raises(ValueError):
            model.forward_pairwise((p[0], p[1], p[1][:, :-1]), (ps[0], ps[1], ps[1]), (sc[0], sc[1], sc[1]))
        T = model.spec.num_tokens
        model.token_num = -T                                 # == token 0
        assert torch.equal(model(p, ps, sc)[0], q)
        model.token_num = -T - 1
        with pytest.raises(RuntimeError, match="vtq_set_iqa_token"):
            model(p, ps, sc)


def test_pos_is_not_looked_at_without_positional_embedding():
    """use_pos_embedding=False: the reference never touches `pos` (transformer.py:539), so None, or coordinates that would index past
    the table, give the same scores as the golden's -- bit for bit, and without the IndexError of a model that has a table."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("nopos_b2_n40")
    model = build(kw, sd, "auto")
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q0, _ = model(p, ps, sc)
        q1, _ = model(p, (None, None), sc)
        q2, _ = model(p, (ps[0] + 7.0, ps[1] - 3.0), sc)
    assert gate(q0.cpu().numpy(), g["q"], TOL["fp16x3"])
    assert torch.equal(q0, q1) and torch.equal(q0, q2)


@pytest.mark.parametrize("precision", ALL_MODES + ["auto"])
@pytest.mark.parametrize("name", FULLSIZE_CASES)
def test_golden_at_the_bench_sizes(name, precision):
    """The whole batch at the sizes bench.py times -- BASELINE configs[1] (B = 32, N = 500, ViT-B/16 L = 12) and the reference-default
    topology (L = 6, 8 registers, LayerScale, r = 16; B = 16, N = 512) -- and BASELINE configs[3] whole (ViT-L/16, L = 24, B = 16,
    N = 1024 over 3 scales), against scores the REFERENCE produced for the same seeded inputs (tests/golden/make_golden.py
    --fullsize / --fullsize-c4).  The parity mode (and `auto`, which must resolve to it) meets the raw per-score gate
    on both.  The throughput modes are gated on the error relative to the batch's rms: the flat-init L = 12 scores of 32 pairs are
    small through cancellation (rms 7e-3, individual scores down to 0.12 rms; both fp32 evaluations are themselves 4e-5 of the rms
    from float64, tests/test_oracle_golden.py), and a raw relative error on such a score measures the cancellation, not the mode."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd, precision)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q, _ = model(p, ps, sc)
    q = q.cpu().numpy()
    e = rel_err(q, g["q"])
    print(f"\n[{name} {precision}] {e}")
    assert q.shape == (int(g["B"]),) and np.isfinite(q).all()
    if precision in ("fp16x3", "auto"):
        assert gate(q, g["q"], TOL["fp16x3"]), e
        if precision == "auto":             # no overflow on these weights: the default model's scores ARE the parity mode's, bit for bit
            with torch.no_grad():
                q3 = build(kw, sd, "fp16x3")(p, ps, sc)[0].cpu().numpy()
            assert model.engine_precision == "fp16x3" and np.array_equal(q.view(np.uint32), q3.view(np.uint32))
    else:
        assert e["max_rel_rms"] < TOL[precision], e


@pytest.mark.parametrize("precision", ["fp16x3", "auto", "bf16x3", "fp16x2", "fp16"])
@pytest.mark.parametrize("name", LONG_CASES)
def test_golden_long_sequence(name, precision):
    """The long-sequence regime the reference advertises (README.md:85 "50, 500, and 5000 patches", data/patch_sampling.py:450): ViT-B/16 L = 12 at
    N = 5000 (S = 5001 -- 20 query blocks of 256 rows and 79 key tiles per head; attention is > 50 % of the flops; the CLS tail's score
    buffer is 5001 entries) against scores the REFERENCE produced (tests/golden/make_golden.py --long): flat seeded weights (2 pairs) and
    trained-like statistics through a head at its operating point (1 pair; fp32 and float64 of the reference).  Parity mode: RAW relative
    error of every score <= 1e-3, no floor.  The pruned CLS tail and the full last layer must agree, and the whole-batch scores must not
    depend on the attention form."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd, precision)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q = model(p, ps, sc)[0].cpu().numpy()
    e = rel_err(q, g["q"])
    e64 = rel_err(q, g["q64"]) if "q64" in g else None
    print(f"\n[{name} {precision}] vs reference fp32 {e['max_rel']:.2e}" + (f", vs reference fp64 {e64['max_rel']:.2e}" if e64 else ""))
    assert q.shape == (int(g["B"]),) and np.isfinite(q).all()
    if precision in ("fp16x3", "auto"):
        assert e["max_rel"] < TOL["fp16x3"], e
        if e64:
            assert e64["max_rel"] < TOL["fp16x3"], e64
    else:
        assert e["max_rel"] < {"bf16x3": 3e-3, "fp16x2": 5e-2, "fp16": 2.5e-1}[precision], e
    if precision == "fp16x3":
        with torch.no_grad():
            q_full = build(kw, sd, precision, engine_options=_lib.OPT_FULL_LAST_LAYER)(p, ps, sc)[0].cpu().numpy()
        assert rel_err(q_full, g["q"])["max_rel"] < TOL["fp16x3"] and rel_err(q_full, q)["max_rel"] < 2e-4
        lib = _lib.load()
        try:
            _lib.check(lib.vtq_debug_attention_variant(0))            # the 4-wave kernel instead of the pipelined one: the same bits
            with torch.no_grad():
                q0 = build(kw, sd, precision)(p, ps, sc)[0].cpu().numpy()
        finally:
            _lib.check(lib.vtq_debug_attention_variant(-1))
        assert np.array_equal(q0.view(np.uint32), q.view(np.uint32))


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("name", STRESS_CASES)
def test_golden_trained_like_statistics(name, precision):
    """Goldens captured from the REFERENCE on stress_state weights (peaked softmax, outlier channels; qk = 3 and 5): the parity
    claim off the flat random init, pinned by the reference itself.  Gate: RAW per-score relative error against
    the reference's fp32 scores, 1e-3 for the parity mode (the reference's own fp32 evaluation is 1.2e-5 / 9.0e-5 from its float64
    scores on these two cases, tests/test_oracle_golden.py); the float64 scores are reported beside it."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd, precision)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q, _ = model(p, ps, sc)
    q = q.cpu().numpy()
    e, e64 = rel_err(q, g["q"]), rel_err(q, g["q64"])
    print(f"\n[{name} {precision}] vs reference fp32 {e['max_rel']:.2e}, vs reference fp64 {e64['max_rel']:.2e}")
    assert np.isfinite(q).all()
    assert e["max_rel"] < TOL[precision] and e64["max_rel"] < TOL[precision], (e, e64)


@pytest.mark.parametrize("name", LADDER_CASES)
def test_golden_trained_like_ladder_at_baseline_patch_count(name):
    """64 pairs (8 images x 8 distortion strengths) at N = 500 on stress_state(qk = 5) weights, scored by the REFERENCE in fp32 and in
    float64 (tests/golden/make_golden.py --ladder): the trained-like parity tail at the BASELINE patch count, pinned by the reference
    itself.  Gate: 1e-3 on the RAW relative error of every score with |q| >= 0.1 rms (smaller scores against the
    rms: the reference's OWN fp32 is 5.4e-3 raw from its float64 on the score that is 0.9 % of the rms, 2.0e-4 in the gate's measure);
    max and p95 of the raw error are printed."""
    g, kw, spec, sd, (patches, pos, scales) = load_ladder_case(name)
    model = build(kw, sd, MAIN)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    qs = []
    with torch.no_grad():
        for i in range(0, patches.shape[0], 32):
            sl = slice(i, i + 32)
            qs.append(model((p[0][sl], p[1][sl]), (ps[0][sl], ps[1][sl]), (None, None))[0])
    q = torch.cat(qs).cpu().numpy().astype(np.float64)
    assert np.isfinite(q).all()
    for tag in ("q", "q64"):
        ref = g[tag].astype(np.float64)
        raw = np.abs(q - ref) / np.abs(ref)
        big = np.abs(ref) >= 0.1 * np.sqrt(np.mean(ref ** 2))
        ge = gate_error(q, ref)
        print(f"\n[{name} {MAIN}] vs reference {'fp32' if tag == 'q' else 'fp64'}: gate {ge:.2e}; raw max over |q| >= 0.1 rms {raw[big].max():.2e}, "
              f"p95 {np.percentile(raw, 95):.2e}, raw max over all {raw.max():.2e} (|q| = {abs(ref[np.argmax(raw)]):.1e})")
        assert ge < 1e-3, (tag, ge)


@pytest.mark.parametrize("name", OPERATING_POINT_CASES)
def test_golden_at_a_trained_models_operating_point(name):
    """The 64-pair N = 500 ladder on stress_state(qk = 5) weights through a head at a trained model's OPERATING POINT (scores in [0.2, 0.8],
    std 0.1: the released checkpoint predicts normalised MOS of that order, data/patch_datasets.py:51-52; every other reference-pinned case
    has scores that are cancellation remainders around zero), scored by the REFERENCE in fp32 and float64 (make_golden.py --operating-point).
    Gate for the parity mode: 1e-3 RAW relative on EVERY score, no rms floor.  The other modes are printed with the same measure and
    gated loosely: here the question is which of them a calibration-checked `auto` could pick (profiles/r05_operating_point_errors.txt)."""
    g, kw, spec, sd, (patches, pos, scales) = load_ladder_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    ref32, ref64 = g["q"].astype(np.float64), g["q64"].astype(np.float64)
    print(f"\n[{name}] reference scores: min {ref64.min():.3f} max {ref64.max():.3f} std {ref64.std():.3f}; reference fp32 vs its own fp64: "
          f"{np.max(np.abs(ref32 - ref64) / np.abs(ref64)):.2e}")
    # parity mode: the north-star tolerance on every score.  The others: smoke bounds only (measured 4.2e-4 / 1.2e-2 / ... : see the profile) --
    # at the operating point, too, nothing cheaper than three MFMAs per product is within 1e-3, bf16x3 excepted (on THESE weights)
    loose = {"fp16x3": 1e-3, "bf16x3": 3e-3, "fp16x2": 1e-1, "fp16": 1.0, "bf16": 3.0}
    for precision in ALL_MODES:
        model = build(kw, sd, precision)
        qs = []
        with torch.no_grad():
            for i in range(0, patches.shape[0], 32):
                sl = slice(i, i + 32)
                qs.append(model((p[0][sl], p[1][sl]), (ps[0][sl], ps[1][sl]), (None, None))[0])
        q = torch.cat(qs).cpu().numpy().astype(np.float64)
        assert np.isfinite(q).all()
        raw32, raw64 = np.abs(q - ref32) / np.abs(ref32), np.abs(q - ref64) / np.abs(ref64)
        print(f"[{name} {precision}] raw relative error of every score: vs reference fp32 max {raw32.max():.2e} p95 {np.percentile(raw32, 95):.2e}; "
              f"vs reference fp64 max {raw64.max():.2e} p95 {np.percentile(raw64, 95):.2e}; max abs {np.abs(q - ref64).max():.2e}")
        assert raw32.max() < loose[precision] and raw64.max() < loose[precision], (precision, raw32.max(), raw64.max())
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ALL_MODES)
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
    tol = {"fp16x3": 2e-5, "fp16x2": 1e-3, "bf16x3": 2e-4, "fp16": 5e-3, "bf16": 4e-2}[precision]
    for layer in range(L):
        d = np.abs(got[layer + 1] - want[layer]).max() / np.abs(want[layer]).max()
        assert d < tol, (layer, d)


def test_plumbing_c1():
    """BASELINE config 1 through the train.py-equivalent boundary: collated batch -> get_data_tuple -> predict."""
    import os
    from tests.helpers import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "plumbing_c1.npz")))
    kw = json.loads(str(g["kwargs"]))
    model = VTAMIQ(**json.loads(json.dumps(kw)), precision=MAIN)
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


@pytest.mark.parametrize("precision", ALL_MODES)
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
        # (4) a patch set is a SET (get_iqa_patches draws the patches in random order, data/patch_sampling.py:529-611; nothing in the
        #     forward depends on the order but the order of the softmax / PV sums): permuting the N patches of every image, with
        #     their positions, moves the score by rounding only -- this crosses every key tile and query block of the attention kernels
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(7)).to(DEV)
        q_perm = model(tuple(t[:, perm].contiguous() for t in p), tuple(t[:, perm].contiguous() for t in ps), (None, None))[0]
        e_perm = gate_error(q_perm.cpu().numpy(), q.cpu().numpy())
        print(f"\n[full-size {precision}] patch order: {e_perm:.2e}")
        # 3-term attention: 2e-5 (fp16x3) .. 2e-4 (bf16x3); single-plane attention rounds P and the scores' operands to 11 / 8 bits,
        # differently per order: the change is the mode's own error level (1.9e-2 fp16, 2.5e-1 bf16)
        assert e_perm < TOL[precision] * (0.2 if precision.endswith(("x3", "x2")) else 1.2), e_perm
    zero = torch.zeros(1, 1, spec.hidden_size)
    q0 = O.head(O.to_torch(sd), spec, zero, zero)
    assert abs(float(q_same[0]) - float(q0[0])) < 1e-6 + 1e-5 * abs(float(q0[0]))
    # (5) oracle on the host for a few pairs of the full-size batch (exact same inputs)
    cp, cps, _ = split_inputs(patches[sel], pos[sel], None)
    q_ref = O.vtamiq_forward(O.to_torch(sd), spec, cp, cps, (None, None))[0].numpy()
    e = rel_err(q[sel].cpu().numpy(), q_ref)
    print(f"\n[full-size {precision}] {e}")
    assert gate(q[sel].cpu().numpy(), q_ref, TOL[precision]), e


@pytest.mark.parametrize("name", ["c1_b2_n50", "refdefault_b2_n64", "vitl_b2_n70"])
def test_cls_pruned_last_layer_matches_full_layer(name):
    """The CLS-only tail of the last layer (cls_tail.hip) against running the full last layer (vtq_config.options & VTQ_OPT_FULL_LAST_LAYER)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q_pruned = build(kw, sd, MAIN)(p, ps, sc)[0].cpu().numpy()
        q_full = build(kw, sd, MAIN, engine_options=_lib.OPT_FULL_LAST_LAYER)(p, ps, sc)[0].cpu().numpy()
    assert gate(q_pruned, g["q"], 1e-3) and gate(q_full, g["q"], 1e-3)
    assert gate(q_pruned, q_full, 3e-4), rel_err(q_pruned, q_full)


@pytest.mark.parametrize("precision", ["fp16x3", "fp16x2", "bf16"])
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
                                                      (1, 130, 0, "ViT-L16", 3), (7, 56, 3, "ViT-B16", 2), (2, 119, 8, "ViT-B16", 0),
                                                      (3, 301, 2, "ViT-B8", 3), (1, 17, 0, "ViT-B8", 0)])
@pytest.mark.parametrize("precision", ["fp16x3", "fp16x2"])
def test_ragged_shapes_against_oracle(B, N, extra, variant, scales, precision):
    """Edge shapes: B = 1 / odd B, S = N + T hitting 9 / 78 / 209 / 128 exactly, register tokens, 2- and 3-scale embeddings,
    ViT-L, ViT-B/8 (8x8 patches: a 192-wide patch embedding padded to the GEMM's K tile, 48 x 48 position grid); oracle on the host
    as the checker."""
    kw = dict(vit_config=dict(variant=variant, num_keep_layers=2, num_extra_tokens=extra, num_scales=scales, use_layer_scale=bool(extra)),
              num_rgs=2, num_rcabs=2, ca_reduction=16)
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
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
    assert gate(q, q_ref, TOL[precision]), rel_err(q, q_ref)


@pytest.mark.parametrize("qk", [3.0, 5.0, 8.0])
def test_trained_like_statistics_against_oracle(qk):
    """The flat random init makes attention uniform and activations small, which flatters reduced-precision operands.  Same
    topology (ViT-B/16, all 12 layers) with peaked softmax rows (mean max-probability ~0.6 / ~0.85 / ~0.95) and outlier channels
    of ~30x the stream's rms.  Near one-hot attention makes the MODEL ill-conditioned: the oracle itself moves by `cond` between
    fp32 and fp64 arithmetic (5e-6 / 2.5e-4 / 5e-4 here), i.e. it amplifies a 6e-8 rounding by up to 1e4.  The bound asked of the
    parity mode (fp16x3, operands to 2^-22) is max(1e-3, 10 * cond): the north-star 1e-3 wherever the fp32 reference is itself
    reproducible to 1e-4, proportionally more where it is not; bf16x3 (2^-17) gets 30 * cond as in round 1.  The other modes' errors
    are printed."""
    kw = dict(vit_config=dict(variant="ViT-B16"))
    spec = VTAMIQ(**json.loads(json.dumps(kw))).spec
    sd = stress_state(spec, 5, qk=qk)
    patches, pos, sc = synth.make_inputs(spec, 3, 90, 9)
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(O.to_torch(sd), spec, cp, cps, cs)[0].numpy()
    q_f64 = O.vtamiq_forward(O.to_torch(sd, dtype=torch.float64), spec, [t.double() for t in cp], [t.double() for t in cps], cs)[0].numpy()
    cond = rel_err(q_ref, q_f64)["max_rel_rms"]
    errs = {}
    for precision in ALL_MODES:
        m = build(kw, sd, precision)
        with torch.no_grad():
            q = m(p, ps, s3)[0].cpu().numpy()
        assert np.isfinite(q).all()
        errs[precision] = rel_err(q, q_ref)["max_rel_rms"]
        if precision in ("fp16x3", "bf16x3"):
            tol = max(1e-3, (10.0 if precision == "fp16x3" else 30.0) * cond)
            assert rel_err(q, q_ref)["max_rel_rms"] < tol, (q, q_ref, errs, cond)
    print("trained-like statistics qk=%g: oracle fp32-vs-fp64 %.1e, errors %s" % (qk, cond, errs))


# ---- evidence for the configurations and rows the golden fixtures only cover at reduced size ------------------------------

def test_config4_full_size_vit_l():
    """BASELINE configs[3] at FULL size: ViT-L/16 (H=1024, L=24, 16 heads), B=16 pairs, N=1024 patches over 3 scales (S=1025:
    17 key tiles, 9 query blocks per sequence).  Batch invariance (bitwise), determinism, and one pair against the oracle."""
    kw = dict(vit_config=dict(variant="ViT-L16", num_scales=3))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=MAIN)
    spec = m.spec
    assert (spec.hidden_size, spec.num_layers, spec.num_heads, spec.mlp_dim) == (1024, 24, 16, 4096)
    sd = synth.make_state_dict(spec, 77)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    B, N = 16, 1024
    patches, pos, sc = synth.make_inputs(spec, B, N, 78)
    assert sc is not None and set(np.unique(sc)) == {0, 1, 2}
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    with torch.no_grad():
        q = m(p, ps, s3)[0]
        sel = [0, 7, 15]
        q_small = m(tuple(t[sel].contiguous() for t in p), tuple(t[sel].contiguous() for t in ps), tuple(t[sel].contiguous() for t in s3))[0]
        assert torch.equal(q[sel], q_small)
        assert torch.equal(q, m(p, ps, s3)[0])
    assert q.shape == (B,) and bool(torch.isfinite(q).all())
    one = [7]
    cp, cps, cs = split_inputs(patches[one], pos[one], sc[one])
    torch.set_num_threads(16)
    q_ref = O.vtamiq_forward(O.to_torch(sd), spec, cp, cps, cs)[0].numpy()
    e = rel_err(q[one].cpu().numpy(), q_ref)
    print(f"\n[configs[3] full size, {MAIN}] q={q[one].cpu().numpy()} ref={q_ref} {e}")
    assert e["max_rel"] < 1e-3, e


@pytest.mark.parametrize("precision", ["fp16x3", "fp16x2"])
def test_long_sequence_beyond_2048_tokens(precision):
    """N = 2500 patches per image (the reference's README advertises up to 5000): S = 2501 goes through the CLS-pruned last layer
    with its score buffer in dynamic LDS (round 1 failed for S > 2048).  2 layers keep the host-side oracle affordable."""
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=2))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    sd = synth.make_state_dict(m.spec, 91)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    patches, pos, sc = synth.make_inputs(m.spec, 2, 2500, 92)
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    with torch.no_grad():
        q = m(p, ps, s3)[0].cpu().numpy()
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(O.to_torch(sd), m.spec, cp, cps, cs)[0].numpy()
    assert gate(q, q_ref, TOL[precision]), rel_err(q, q_ref)


def test_ingested_npz_weights_through_the_engine(tmp_path):
    """SURVEY 8f-3 on the GPU: a JAX-layout ViT-B/16 .npz (synthetic, seeded; 14x14 position grid so the bilinear resize to 24x24
    runs) -> weights.load_vit_npz -> engine scores, against the oracle on the state dict the (CPU-pinned) converter returns."""
    from vtamiq_amd import weights as W
    L = 2
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=L, pretrained=False), num_rgs=1, num_rcabs=2)
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=MAIN)
    rs = np.random.RandomState(11)
    H, M, P = 768, 3072, 16
    n = lambda *s: (0.02 * rs.randn(*s)).astype(np.float32)
    w = {"embedding/kernel": n(P, P, 3, H), "embedding/bias": n(H), "cls": n(1, 1, H),
         "Transformer/posembed_input/pos_embedding": n(1, 14 * 14 + 1, H), "Transformer/encoder_norm/scale": 1 + n(H),
         "Transformer/encoder_norm/bias": n(H)}
    for i in range(L):
        r = f"Transformer/encoderblock_{i}"
        for nm in ("query", "key", "value"):
            w[f"{r}/MultiHeadDotProductAttention_1/{nm}/kernel"] = n(H, 12, 64)
            w[f"{r}/MultiHeadDotProductAttention_1/{nm}/bias"] = n(12, 64)
        w[f"{r}/MultiHeadDotProductAttention_1/out/kernel"] = n(12, 64, H)
        w[f"{r}/MultiHeadDotProductAttention_1/out/bias"] = n(H)
        w[f"{r}/MlpBlock_3/Dense_0/kernel"] = n(H, M); w[f"{r}/MlpBlock_3/Dense_0/bias"] = n(M)
        w[f"{r}/MlpBlock_3/Dense_1/kernel"] = n(M, H); w[f"{r}/MlpBlock_3/Dense_1/bias"] = n(H)
        for ln in ("LayerNorm_0", "LayerNorm_2"):
            w[f"{r}/{ln}/scale"] = 1 + n(H); w[f"{r}/{ln}/bias"] = n(H)
    path = tmp_path / "vit_b16.npz"
    np.savez(path, **w)
    m = m.to(DEV).eval()
    patches, pos, sc = synth.make_inputs(m.spec, 3, 40, 12)
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    with torch.no_grad():
        q_before = m(p, ps, s3)[0].clone()         # packs the constructor's random weights first: the reload must be seen
        W.load_vit_npz(m, str(path))
        q = m(p, ps, s3)[0].cpu().numpy()
    assert not np.allclose(q, q_before.cpu().numpy())
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    conv = W.convert_vit_npz(w, H, L, 577)
    for k, v in conv.items():
        assert torch.equal(sd[k].reshape(v.shape), v), k
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(sd, m.spec, cp, cps, cs)[0].numpy()
    assert gate(q, q_ref, 1e-3), rel_err(q, q_ref)


def test_position_out_of_range_is_clamped_and_reported():
    """pos == 1.0, pos < 0 and NaN index outside the 577-row table in the reference (IndexError / device assert,
    transformer.py:417-421).  The engine clamps the index (finite scores, no out-of-bounds gather) and reports it."""
    model, _ = _c2_model("fp16")
    spec = model.spec
    patches, pos, _ = synth.make_inputs(spec, 2, 24, 3)
    p, ps, sc = split_inputs(patches, pos, None, device=DEV)
    with torch.no_grad():
        q_ok = model(p, ps, sc)[0]
        model.check_inputs()                                   # in range: no error
        bad = ps[0].clone()
        bad[0, 0, 0] = 1.0
        bad[0, 1, 1] = -0.25
        bad[1, 2, 0] = float("nan")
        q_bad = model(p, (bad, ps[1]), sc)[0]
        assert bool(torch.isfinite(q_bad).all())
        with pytest.raises(IndexError):
            model.check_inputs()
        model.check_inputs()                                   # the flag was cleared by the read
        model.validate_inputs = True
        with pytest.raises(IndexError):
            model(p, (bad, ps[1]), sc)
        model.validate_inputs = False
        assert torch.equal(model(p, ps, sc)[0], q_ok)


def test_patch_sample_range_is_checked():
    from vtamiq_amd.patches import extract_patches
    img = torch.zeros(1, 64, 80, 3, dtype=torch.uint8, device=DEV)
    ok = torch.tensor([[[0, 0], [48, 64]]], dtype=torch.int32)
    extract_patches(img, ok)
    for bad in ([[49, 0]], [[0, 65]], [[-1, 0]]):
        with pytest.raises(IndexError):
            extract_patches(img, torch.tensor([bad], dtype=torch.int32))
    with pytest.raises(IndexError):
        extract_patches(img, ok, scale_ids=torch.tensor([[0, 2]], dtype=torch.int32), num_scales=2)
    with pytest.raises(ValueError):
        extract_patches(torch.zeros(1, 24, 24, 3, dtype=torch.uint8, device=DEV), torch.zeros(1, 1, 2, dtype=torch.int32), num_scales=2,
                        scale_ids=torch.zeros(1, 1, dtype=torch.int32))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process (the gpurun box has one)")
def test_same_process_two_devices():
    """the library's per-device state (GEMM tile schedules, the > 64 KiB LDS kernel attributes) is keyed by the current
    device: the same model runs on cuda:0 and then on cuda:1 of one process and gives identical scores."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    qs = []
    for dev in ("cuda:0", "cuda:1"):
        m = VTAMIQ(**json.loads(json.dumps(kw)), precision=MAIN)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).eval()
        p, ps, sc = split_inputs(patches, pos, scales, device=dev)
        with torch.cuda.device(dev), torch.no_grad():
            qs.append(m(p, ps, sc)[0].cpu().numpy())
    assert np.array_equal(qs[0], qs[1]) and gate(qs[0], g["q"], TOL[MAIN])


def test_operand_range_overflow_is_reported():
    """fp16 operand planes carry |v| <= 65504: a model whose activations exceed that produces inf / NaN, which reach the CLS rows
    through the softmax; the engine flags a non-finite CLS difference and check_inputs() raises instead of returning NaN silently."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    ok = build(kw, sd, MAIN)
    with torch.no_grad():
        ok(p, ps, sc)
    ok.check_inputs()                                          # a healthy model raises nothing
    bad_sd = {k: v.copy() for k, v in sd.items()}
    bad_sd["transformer.encoder.layers.0.attention_norm.weight"][:4] *= 1e7      # LayerNorm outputs of ~1e7 in four channels
    bad = build(kw, bad_sd, MAIN)
    with torch.no_grad():
        q = bad(p, ps, sc)[0]
    assert not bool(torch.isfinite(q).all())
    with pytest.raises(FloatingPointError):
        bad.check_inputs()
    bad.check_inputs()                                         # the flag is cleared by the check
    # the poisoned workspace does not leak into later healthy forwards of the SAME engine (stale rows behind a smaller batch)
    bad.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    with torch.no_grad():
        q_small = bad(tuple(t[:1] for t in p), tuple(t[:1] for t in ps), sc)[0]
        q_ref_small = ok(tuple(t[:1] for t in p), tuple(t[:1] for t in ps), sc)[0]
    assert torch.equal(q_small, q_ref_small)
    bad.check_inputs()
    strict = build(kw, bad_sd, MAIN)
    strict.validate_inputs = True                              # check after every forward (or VTAMIQ_VALIDATE_INPUTS=1)
    with pytest.raises(FloatingPointError), torch.no_grad():
        strict(p, ps, sc)


def test_default_model_has_no_silent_nans():
    """The DEFAULT-constructed drop-in (precision='auto') on the same 1e7-gain weights: the fp32 reference
    returns finite scores there (train.py:602-607), so must the drop-in, without the user calling anything: the fp16 operand
    overflow is noticed after the forward, the model switches itself to bf16x3 (fp32 operand range) with a warning, repeats the
    call and returns scores within the bf16x3 bound of the fp32 oracle.  A healthy model stays in fp16x3 and equals the explicit mode
    bit for bit; a position outside [0, 1) raises the reference's IndexError."""
    import warnings
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    auto = VTAMIQ(**json.loads(json.dumps(kw)))                 # no precision argument
    assert auto.precision == "auto"
    auto.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    auto = auto.to(DEV).eval()
    with torch.no_grad():
        q_auto = auto(p, ps, sc)[0]
        q_x3 = build(kw, sd, "fp16x3")(p, ps, sc)[0]
    assert auto.engine_precision == "fp16x3" and torch.equal(q_auto, q_x3)
    bad_sd = {k: v.copy() for k, v in sd.items()}
    bad_sd["transformer.encoder.layers.0.attention_norm.weight"][:4] *= 1e7
    bad = VTAMIQ(**json.loads(json.dumps(kw)))
    bad.load_state_dict({k: torch.from_numpy(v) for k, v in bad_sd.items()})
    bad = bad.to(DEV).eval()
    with warnings.catch_warnings(record=True) as w, torch.no_grad():
        warnings.simplefilter("always")
        q = bad(p, ps, sc)[0]
    assert any("bf16x3" in str(x.message) for x in w)
    assert bad.engine_precision == "bf16x3" and bool(torch.isfinite(q).all())
    q_ref = O.vtamiq_forward(O.to_torch(bad_sd), spec, *split_inputs(patches, pos, scales))[0].numpy()
    e = rel_err(q.cpu().numpy(), q_ref)
    print(f"\n[auto -> bf16x3 on the 1e7-gain weights] {e}")
    assert gate(q.cpu().numpy(), q_ref, TOL["bf16x3"]), e
    with torch.no_grad():                                       # sticky: the next call runs bf16x3 directly, same scores
        assert torch.equal(bad(p, ps, sc)[0], q)
    oob = tuple(t.clone() for t in ps)
    oob[0][0, 3, 1] = 1.0                                       # pos == 1.0: one past the table (transformer.py:417-421)
    with pytest.raises(IndexError), torch.no_grad():
        auto(p, oob, sc)


def test_default_model_nan_inputs_do_not_change_its_mode():
    """bit 1 of the error word is also raised by inf / NaN INPUTS.  One bad batch must not leave the default model in
    bf16x3 for good: the bf16x3 re-run is non-finite as well, so the model goes back to fp16x3, warns, and returns what the reference's
    fp32 forward returns for such a batch -- NaN scores for the pairs that hold the NaN (train.py:602-607), no exception."""
    import warnings
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    auto = VTAMIQ(**json.loads(json.dumps(kw)))
    auto.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    auto = auto.to(DEV).eval()
    with torch.no_grad():
        q_good = auto(p, ps, sc)[0]
    bad_p = (p[0].clone(), p[1].clone())
    bad_p[1][1, 7, 2, 3, 3] = float("nan")                      # one NaN sample in pair 1's distorted image
    with warnings.catch_warnings(record=True) as w, torch.no_grad():
        warnings.simplefilter("always")
        q = auto(bad_p, ps, sc)[0]
    assert any("inf / NaN" in str(x.message) for x in w)
    assert auto.engine_precision == "fp16x3"                    # not sticky
    qr = O.vtamiq_forward(O.to_torch(sd), spec, (bad_p[0].cpu(), bad_p[1].cpu()), (ps[0].cpu(), ps[1].cpu()), (None, None))[0]
    assert torch.equal(torch.isnan(q).cpu(), torch.isnan(qr))   # NaN exactly where the fp32 reference path has it (the attention kernels zero
                                                                # the V rows of a sequence's masked keys: they belong to the NEXT sequence)
    assert torch.equal(q[0], q_good[0])                         # and the healthy pair of the batch scores the same bits as before
    with torch.no_grad():
        assert torch.equal(auto(p, ps, sc)[0], q_good)          # and the next healthy batch scores as before, in the parity mode
    # the probe's bf16x3 engine is parked with its packed weights, not destroyed: a second bad batch builds and re-packs nothing
    parked = auto.__dict__["_parked"]
    assert "bf16x3" in parked and parked["bf16x3"][1] is not None
    h_bf16, h_fp16 = parked["bf16x3"][0].value, auto._engine.value
    with warnings.catch_warnings(record=True), torch.no_grad():
        warnings.simplefilter("always")
        q2 = auto(bad_p, ps, sc)[0]
    assert torch.equal(torch.isnan(q2), torch.isnan(q)) and torch.equal(q2[0], q_good[0])
    assert auto.__dict__["_parked"]["bf16x3"][0].value == h_bf16 and auto._engine.value == h_fp16 and auto.engine_precision == "fp16x3"


def test_library_ignores_measurement_environment(monkeypatch):
    """the product library reads no environment variable.  With every former knob set to its most destructive
    value (VTQ_GEMM_FLAGS=8 used to skip the GEMM epilogues) a NEW process-independent engine still returns the golden scores.
    (The library was loaded long before this test: the knobs used to be read lazily at first launch / engine creation, so the
    engine-level ones -- and a fresh GEMM shape's schedule -- would still have seen them.)"""
    for k, v in (("VTQ_GEMM_FLAGS", "8"), ("VTQ_GEMM_CUS", "3"), ("VTQ_GEMM_SCHED", "2"), ("VTQ_GEMM_CG", "1"), ("VTQ_GEMM_STAGGER", "0"),
                 ("VTQ_NO_CLS_PRUNE", "1"), ("VTQ_ATTN_VARIANT", "0"), ("VTQ_ATTN_LDS_PAD", "65536")):
        monkeypatch.setenv(k, v)
    import subprocess, sys
    code = ("import json, numpy as np, torch; from tests.helpers import load_case, split_inputs, gate_error; from vtamiq_amd import VTAMIQ;"
            "g, kw, spec, sd, (pa, po, sc) = load_case('unaligned_b3_n50');"
            "m = VTAMIQ(**json.loads(json.dumps(kw)), precision='fp16x3'); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()});"
            "m = m.cuda().eval(); p, ps, s = split_inputs(pa, po, sc, device='cuda');"
            "q = m(p, ps, s)[0].cpu().numpy(); print('ERR', gate_error(q, g['q']))")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    err = float(r.stdout.strip().split("ERR")[-1])
    assert err < 1e-3, err


@pytest.mark.parametrize("name,precision", [("c2shape_b4_n500", "fp16x3"), ("vitl_b2_n70", "fp16x3"), ("scales3_b2_n40", "bf16x3"),
                                            ("refdefault_b2_n64", "fp16x2"), ("c2shape_b4_n500", "fp16")])
def test_attention_kernels_agree_end_to_end(name, precision):
    """The engine's own rule sends small batches to the 4-wave attention kernel and chip-filling ones to the software-pipelined kernel
    (attention.hip use_pipelined); forced either way the scores of a forward are the same bits -- on goldens with extra tokens, three
    scales, ViT-L and the single-plane format the rule never gives to the pipelined kernel -- and 12 repeated forwards through the
    pipelined kernel (persistent workgroups, LDS ring, counted waits: a race would show) repeat bit for bit."""
    from vtamiq_amd import _lib
    lib = _lib.load()
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    got = []
    for variant in (0, 1):
        lib.vtq_debug_attention_variant(variant)
        try:
            model = build(kw, sd, precision)
            with torch.no_grad():
                q = model(p, ps, sc)[0]
                for _ in range(12 if variant == 1 else 1):
                    assert torch.equal(model(p, ps, sc)[0], q)
            got.append(q.cpu().numpy())
        finally:
            lib.vtq_debug_attention_variant(-1)
    assert np.array_equal(got[0], got[1])
    assert gate(got[1], g["q"], TOL[precision]), rel_err(got[1], g["q"])


@pytest.mark.parametrize("name,precision", [("c2shape_b4_n500", "fp16x3"), ("vitl_b2_n70", "fp16x3"), ("scales3_b2_n40", "bf16x3"),
                                            ("adapters_b2_n40", "fp16x3"), ("refdefault_b2_n64", "fp16x2"), ("vitb8_b2_n90", "fp16"), ("c1_b2_n50", "bf16")])
def test_gemm_tile_shapes_agree_end_to_end(name, precision):
    """The library's rule gives small batches the small-tile GEMM kernels (csrc/gemm_st.hip) and chip-filling ones the persistent
    256x256 kernel; forced to any one shape -- patch embedding, QKV, out-proj, fc1, fc2 and the adapter GEMMs alike -- a forward's scores
    are the same bits (the bitwise contract that makes the choice a pure speed choice), and 8 repeated forwards per shape repeat."""
    lib = _lib.load()
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    got = {}
    for variant in (-1, 0, 1, 2, 3):
        _lib.check(lib.vtq_debug_gemm_variant(variant))
        try:
            model = build(kw, sd, precision)
            with torch.no_grad():
                q = model(p, ps, sc)[0]
                for _ in range(8):
                    assert torch.equal(model(p, ps, sc)[0], q)
            got[variant] = q.cpu().numpy()
        finally:
            _lib.check(lib.vtq_debug_gemm_variant(-1))
    for variant, q in got.items():
        assert np.array_equal(q, got[0]), variant
    assert gate(got[-1], g["q"], TOL[precision]), rel_err(got[-1], g["q"])


@pytest.mark.parametrize("precision", ["fp16x3", "fp16"])
def test_repeated_forwards_are_bitwise_identical(precision):
    """Race detector for the persistent GEMM (DMA ring chained across tile boundaries, counted vmcnt, raw barriers) and every other
    kernel: the same batch 25 times -- interleaved with a different batch that reuses the workspace -- gives bit-identical scores."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c2shape_b4_n500")
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    other = tuple(t.flip(0).contiguous() for t in p)
    with torch.no_grad():
        q0 = m(p, ps, sc)[0].clone()
        for i in range(25):
            if i % 3 == 0:
                m(other, ps, sc)
            q = m(p, ps, sc)[0]
            assert torch.equal(q, q0), (i, (q - q0).abs().max().item())


def test_workspace_regrowth_keeps_results():
    """The workspace is sized by the high-water mark of (B, N): small -> large -> small -> longer sequences on ONE engine give the
    scores a fresh engine gives for each call (no stale rows, padding or schedules carried over)."""
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=2))
    spec = VTAMIQ(**json.loads(json.dumps(kw))).spec
    sd = synth.make_state_dict(spec, 77)
    one = build(kw, sd, MAIN)
    for B, N, seed in [(2, 40, 1), (19, 130, 2), (3, 40, 3), (1, 700, 4), (5, 9, 5)]:
        patches, pos, sc = synth.make_inputs(spec, B, N, seed)
        p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
        fresh = build(kw, sd, MAIN)
        with torch.no_grad():
            qa, qb = one(p, ps, s3)[0], fresh(p, ps, s3)[0]
        assert torch.equal(qa, qb), (B, N, (qa - qb).abs().max().item())
        del fresh


@pytest.mark.parametrize("variant,precision", [("ViT-B16", "fp16x3"), ("ViT-L16", "fp16x3"), ("ViT-B16", "fp16x2"), ("ViT-B8", "bf16x3")])
def test_adapters_against_oracle(variant, precision):
    """Adapter pair 0 after attention and after the MLP (transformer.py:177-194, 279-283): H/4 = 192 padded to the GEMM tile
    (ViT-B) or 256 as is (ViT-L); other pairs are accepted and ignored like in the reference's forward; pairwise entry too."""
    kw = dict(vit_config=dict(variant=variant, num_keep_layers=2, num_adapters=3, use_layer_scale=True, num_scales=2), num_rgs=1, num_rcabs=2)
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    sd = synth.make_state_dict(m.spec, 91)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    patches, pos, sc = synth.make_inputs(m.spec, 3, 77, 92)
    p, ps, s3 = split_inputs(patches, pos, sc, device=DEV)
    with torch.no_grad():
        q = m(p, ps, s3)[0]
        f1, f2 = m.forward_pairwise((p[0], p[1], p[1]), (ps[0], ps[1], ps[1]), (s3[0], s3[1], s3[1]))
    cp, cps, cs = split_inputs(patches, pos, sc)
    q_ref = O.vtamiq_forward(O.to_torch(sd), m.spec, cp, cps, cs)[0].numpy()
    assert gate(q.cpu().numpy(), q_ref, TOL[precision]), rel_err(q.cpu().numpy(), q_ref)
    assert torch.equal(f1, q) and torch.equal(f2, q)
    # the adapters matter: zeroing pair 0's up-projection changes the scores, zeroing pair 1's does not
    sd2 = {k: v.copy() for k, v in sd.items()}
    for k in sd2:
        if ".adapter3." in k or ".adapter4." in k:
            sd2[k][...] = 0
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    with torch.no_grad():
        q_same = m(p, ps, s3)[0]
    for k in sd2:
        if ".adapter1.adapter.2." in k:
            sd2[k][...] = 0
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    with torch.no_grad():
        q_diff = m(p, ps, s3)[0]
    assert torch.equal(q_same, q) and not torch.equal(q_diff, q)


def test_rccl_single_rank_all_gather_of_the_scores():
    """the N > 1 path's one collective, executed on hardware with ONE rank -- init_process_group('nccl') (= RCCL),
    the engine's forward and vtamiq_amd.dist.gather_scores -> all_gather_into_tensor on the compute stream, in a child process
    (the process group must not leak into the test session).  Scores through the collective equal the direct ones bit for bit."""
    import os, subprocess, sys
    code = ("import json, os, torch, torch.distributed as dist; from tests.helpers import load_case, split_inputs; from vtamiq_amd import VTAMIQ;"
            "from vtamiq_amd.dist import gather_scores;"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29731'); torch.cuda.set_device(0);"
            "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0));"
            "g, kw, spec, sd, (pa, po, sc) = load_case('c1_b2_n50');"
            "m = VTAMIQ(**json.loads(json.dumps(kw)), precision='fp16x3'); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()});"
            "m = m.cuda().eval(); p, ps, s = split_inputs(pa, po, sc, device='cuda');"
            "q0 = m(p, ps, s)[0]; q1 = gather_scores(m(p, ps, s)[0], q0.numel(), force_collective=True); torch.cuda.synchronize();"
            "print('SAME', int(torch.equal(q0, q1) and q1.data_ptr() != q0.data_ptr()), dist.get_backend()); dist.destroy_process_group()")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SAME 1 nccl" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("name,precision", [("c1_b2_n50", "fp16x3"), ("refdefault_b2_n64", "fp16x3"), ("unaligned_b3_n50", "bf16x3"), ("c2shape_b4_n500", "fp16x3"),
                                            ("nocalib_b2_n30", "fp16x3"), ("stress5_b3_n90", "fp16x3")])
def test_layernorm_inside_the_residual_gemms_is_bit_identical(name, precision):
    """vtq_config.options & VTQ_OPT_FUSED_LAYERNORM (csrc/gemm_rowln.hip: the out-proj launch writes LayerNorm 2's planes, the fc2 launch
    the next layer's LayerNorm 1 planes, 22 of the 23 LayerNorm launches of a ViT-B forward disappear): the same accumulation order and
    the same LayerNorm arithmetic as the separate launches, so the SCORES are the same bits -- with registers, LayerScale, the CLS-pruned
    and the full last layer, one layer (no LayerNorm behind its fc2), trained-like weights -- and every per-layer token row of the trace."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    B, L, T, H = int(g["B"]), spec.num_layers, spec.num_tokens, spec.hidden_size
    with torch.no_grad():
        q_sep = build(kw, sd, precision)(p, ps, sc)[0]
        q_fused = build(kw, sd, precision, engine_options=_lib.OPT_FUSED_LAYERNORM)(p, ps, sc)[0]
        q_full = build(kw, sd, precision, engine_options=_lib.OPT_FUSED_LAYERNORM | _lib.OPT_FULL_LAST_LAYER)(p, ps, sc)[0]
        q_full_sep = build(kw, sd, precision, engine_options=_lib.OPT_FULL_LAST_LAYER)(p, ps, sc)[0]
        ta, tb = torch.zeros(L + 1, 2 * B, T, H, device=DEV), torch.zeros(L + 1, 2 * B, T, H, device=DEV)
        build(kw, sd, precision)(p, ps, sc, _trace=ta)
        build(kw, sd, precision, engine_options=_lib.OPT_FUSED_LAYERNORM)(p, ps, sc, _trace=tb)
    assert torch.equal(q_sep, q_fused) and torch.equal(q_full, q_full_sep)
    assert torch.equal(ta, tb)
    assert gate(q_fused.cpu().numpy(), g["q"], TOL[precision])


def test_fused_layernorm_option_is_rejected_where_the_kernel_does_not_apply():
    for kw, precision in ((dict(vit_config=dict(variant="ViT-L16", num_keep_layers=1, pretrained=False)), "fp16x3"),
                          (dict(vit_config=dict(variant="ViT-B16", num_keep_layers=1, pretrained=False)), "fp16"),
                          (dict(vit_config=dict(variant="ViT-B16", num_keep_layers=1, num_adapters=1, pretrained=False)), "fp16x3")):
        m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision, engine_options=_lib.OPT_FUSED_LAYERNORM).to(DEV).eval()
        N = 16
        p = torch.zeros(1, N, 3, 16, 16, device=DEV)
        pos = torch.zeros(1, N, 2, device=DEV)
        sc = (torch.zeros(1, N, device=DEV),) * 2 if m.spec.use_scale_embedding else (None, None)
        with pytest.raises(RuntimeError, match="VTQ_OPT_FUSED_LAYERNORM"), torch.no_grad():
            m((p, p), (pos, pos), sc)


@pytest.mark.parametrize("nscales", [1, 3])
def test_image_pair_pipeline_matches_the_direct_path(nscales):
    """vtamiq_amd.pipeline.ImagePairPipeline (pinned host buffers -> H2D on a copy stream -> on-device gather -> forward, two buffer
    sets) over five batches -- the buffer sets wrap around twice -- against extract_patches + the model called directly on the same
    images and samples: the same bits.  An out-of-range sample raises the loader's IndexError before anything is enqueued."""
    from vtamiq_amd.patches import extract_patches
    from vtamiq_amd.pipeline import ImagePairPipeline
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=2, num_scales=nscales if nscales > 1 else 0, pretrained=False))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=MAIN)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(m.spec, 3).items()})
    m = m.to(DEV).eval()
    B, N, H, W = 3, 60, 96, 128
    rs = np.random.RandomState(5)
    pipe = ImagePairPipeline(m, B, (H, W), N, num_scales=nscales)
    counts = synth.num_patches_per_scale(N, nscales) if nscales > 1 else np.array([N])
    sid1 = np.concatenate([np.full(c, s) for s, c in enumerate(counts)]).astype(np.int32)
    got, want = [], []
    for it in range(5):
        ref = rs.randint(0, 256, size=(B, H, W, 3), dtype=np.uint8)
        dist = rs.randint(0, 256, size=(B, H, W, 3), dtype=np.uint8)
        smp = np.zeros((B, N, 2), np.int32)
        for s in range(nscales):
            sel = sid1 == s
            smp[:, sel, 0] = rs.randint(0, (H >> s) - 15, size=(B, int(sel.sum())))
            smp[:, sel, 1] = rs.randint(0, (W >> s) - 15, size=(B, int(sel.sum())))
        sid = np.broadcast_to(sid1, (B, N)).copy() if nscales > 1 else None
        got.append(pipe.submit(ref, dist, smp, sid))
        img = torch.from_numpy(np.concatenate([ref, dist])).to(DEV)
        s2 = torch.from_numpy(np.concatenate([smp, smp])).to(DEV)
        i2 = torch.from_numpy(np.concatenate([sid, sid])).to(DEV) if sid is not None else None
        pa, po, sc = extract_patches(img, s2, i2, nscales)
        with torch.no_grad():
            want.append(m((pa[:B], pa[B:]), (po[:B], po[B:]), (sc[:B], sc[B:]) if sc is not None else (None, None))[0])
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    bad = smp.copy()
    bad[0, 0, 0] = H                                        # one row past the image
    with pytest.raises(IndexError):
        pipe.submit(ref, dist, bad, sid)

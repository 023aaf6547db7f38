"""The fake-quant oracle of the fp8 numerics mode (oracle/fp8_oracle.py): its rounding rules and its distance to the fp32 model."""
import os

import numpy as np
import torch

from oracle import fp8_oracle as F8
from oracle import vtamiq_oracle as O
from tests.helpers import GOLDEN, gate_error, load_case, split_inputs


def test_e4m3_grid():
    x = torch.tensor([0.0, 1.0, 1.0625, 1.1875, 447.0, 448.0, 464.0, 1e6, -1e6, 2.0 ** -9, 2.0 ** -10, 0.0146484375])
    q = F8.to_e4m3(x)
    # 3 mantissa bits, ties to even, saturating at 448, subnormals down to 2^-9 kept (7.5 * 2^-9 -> 8 * 2^-9, 2^-10 -> 0: ties to even)
    assert q.tolist() == [0.0, 1.0, 1.0, 1.25, 448.0, 448.0, 448.0, 448.0, -448.0, 2.0 ** -9, 0.0, 0.015625]
    assert torch.equal(F8.to_e4m3(q), q)


def test_row_scales_exact():
    g = torch.Generator().manual_seed(0)
    W = torch.randn(64, 96, generator=g) * torch.logspace(-6, 3, 64)[:, None]
    W[3] = 0
    W[4, 0], W[5, 0], W[6, 0] = 448.0, 448.0 * 4 * (1 + 2 ** -20), 7.0 / 1024      # maxima exactly at / just past a power-of-two boundary
    W[4, 1:], W[5, 1:], W[6, 1:] = 0, 0, 0
    s = F8.row_scales(W)
    m = W.abs().amax(1)
    assert s[3] == 1 and s[4] == 1 and s[5] == 0.125 and s[6] == 65536
    nz = m > 0
    assert (torch.log2(s) == torch.log2(s).round()).all()
    assert (m[nz] * s[nz] <= 448).all() and (m[nz] * s[nz] * 2 > 448).all()
    W8, inv = F8.quant_rows(W)
    assert torch.equal(inv * s, torch.ones_like(s))
    assert ((W8 * inv[:, None] - W).abs() <= 2.0 ** -4 * W.abs() + 2.0 ** -10 * inv[:, None]).all()     # half an ulp of 3 mantissa bits


def test_linear8_is_exact_products():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(8, 64, generator=g, dtype=torch.float64)
    W = torch.randn(16, 64, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(16, generator=g, dtype=torch.float64)
    a8 = F8.quant_act(a, F8.S_LN)
    y = F8.linear8(a8, F8.S_LN, W, b)
    W8, inv = F8.quant_rows(W)
    assert torch.allclose(y, (a8 / F8.S_LN) @ (W8 * inv[:, None]).t() + b, rtol=1e-13, atol=1e-13)
    assert (y - (a @ W.t() + b)).abs().max() < 0.2          # a coarse approximation of the exact layer, as intended


def test_fp8_model_fixture_and_distance():
    """The fp8 model's scores are pinned (tests/golden/fp8_model.npz) and stay a coherent approximation of the fp32 model:
    the distance is tens of percent on these seeded random-init cases (the score is a function of the DIFFERENCE of two CLS rows
    whose 3-mantissa-bit rounding noise is not common-mode) -- the reason this mode is reported beside, not under, the 1e-3 gate."""
    fix = np.load(os.path.join(GOLDEN, "fp8_model.npz"))
    for name in ["c1_b2_n50", "scales3_b2_n40"]:
        g, kw, spec, sd, (patches, pos, scales) = load_case(name)
        p, ps, sc = split_inputs(patches, pos, scales, dtype=torch.float64)
        q = F8.vtamiq_forward(O.to_torch(sd, torch.float64), spec, p, ps, sc)[0].numpy()
        assert np.abs(q - fix[name]).max() < 1e-9
        d = gate_error(q, g["q"])
        print(f"\n[{name}] fp8 model vs fp32 model: {d:.3e}")
        assert 1e-4 < d < 2.0


def test_pick_scale_and_calibrate():
    """The calibration rule (engine.hip fp8_pick_scale): the largest power of two mapping max |value| to <= 224, exact at the
    boundaries; `calibrate` yields scales under which nothing of the calibration batch is clamped."""
    assert F8.pick_scale(224.0, 1.0) == 1.0 and F8.pick_scale(224.0001, 1.0) == 0.5 and F8.pick_scale(112.0, 1.0) == 2.0
    assert F8.pick_scale(1.75, 1.0) == 128.0 and F8.pick_scale(1.7500001, 1.0) == 64.0 and F8.pick_scale(0.0, 8.0) == 8.0
    assert F8.pick_scale(float("nan"), 4.0) == 4.0 and F8.pick_scale(3e-9, 1.0) == 2.0 ** 36
    g, kw, spec, sd, (patches, pos, scales) = load_case("scales3_b2_n40")
    p, ps, sc = split_inputs(patches, pos, scales)
    s8 = F8.calibrate(O.to_torch(sd), spec, p, ps, sc)
    assert len(s8.ln1) == spec.num_layers and s8.patch == F8.pick_scale(float(np.abs(patches).max()), 1.0)
    seen = []

    def pick(name, i, t):
        seen.append(float(t.abs().max()) * getattr(s8, name)[i])
    sdt = O.to_torch(sd)
    x = F8.embeddings(sdt, spec, torch.cat(p), torch.cat(ps), torch.cat(sc), s8)
    for i in range(spec.num_layers):
        x = F8.encoder_layer(sdt, spec, i, x, s8, pick)
    assert len(seen) == 4 * spec.num_layers and max(seen) <= F8.FP8_TARGET and min(seen) > F8.FP8_TARGET / 2.0 - 1e-3
    q_cal = F8.vtamiq_forward(sdt, spec, p, ps, sc, s8=s8)[0]
    q_static = F8.vtamiq_forward(sdt, spec, p, ps, sc)[0]
    assert torch.isfinite(q_cal).all() and q_cal.shape == q_static.shape

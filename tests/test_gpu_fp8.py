"""The fp8 EXPERIMENT (include/vtamiq_hip_fp8.h VTQ_PREC_FP8, vtamiq_amd/experimental_fp8.py; BASELINE.json configs[4]) on the GPU, through the C ABI:
  * the e4m3 rounding of weights and activations is BIT-exact against the fake-quant oracle's rules (oracle/fp8_oracle.py);
  * the MX-scaled-MFMA GEMM reproduces the exact products of the operand bytes (fp64 reference) with each of its epilogues;
  * the engine computes the fp8 MODEL the oracle defines: stage by stage on the engine's own stage inputs (the only form in which
    two implementations of a model with 3-mantissa-bit rounding points can agree), and end to end at the same distance from the
    fp32 model as the oracle's.  That distance is printed (tens of percent on the seeded random-init cases), never gated: this
    mode makes no 1e-3-of-fp32 claim.
"""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from oracle import fp8_oracle as F8
from oracle import vtamiq_oracle as O
from tests.gpu_util import planes_value, stream
from tests.helpers import E2E_CASES, gate_error, load_case, split_inputs
from vtamiq_amd import _lib
from vtamiq_amd.experimental_fp8 import VTAMIQFp8 as VTAMIQ

# The fp8 EXPERIMENT is not in the product library: it is a second build of the engine sources (libvtamiq_hip_fp8.so, made by
# __graft_entry__.build() / python -m vtamiq_amd.build --fp8) that this file reaches through its own handle, _lib.load_fp8() -- so these
# tests run in the default `pytest -m gpu` next to the product library's, in the same process.  A missing library FAILS them (it does not skip).
pytestmark = [pytest.mark.gpu]
DEV = "cuda"


def _chk(rc):
    _lib.check(rc, _lib.load_fp8())


def _randn(*s, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


def e4m3_values(b: torch.Tensor) -> torch.Tensor:
    return b.view(torch.float8_e4m3fn).double()


def quant_rows_gpu(W):
    lib = _lib.load_fp8()
    N, K = W.shape
    w8 = torch.empty(N, K, dtype=torch.uint8, device=DEV)
    inv = torch.empty(N, dtype=torch.float32, device=DEV)
    _chk(lib.vtq_k_quant_rows_fp8(W.data_ptr(), w8.data_ptr(), inv.data_ptr(), N, K, stream()))
    return w8, inv


def quant_act_gpu(x, scale):
    lib = _lib.load_fp8()
    out = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    _chk(lib.vtq_k_quant_fp8(x.data_ptr(), out.data_ptr(), x.numel(), scale, stream()))
    return out


def test_quant_rows_bit_exact():
    W = _randn(777, 768, seed=1) * torch.logspace(-7, 2.5, 777, device=DEV)[:, None]
    W[5] = 0
    W[6, :] = 0; W[6, 3] = 448.0                     # maxima exactly on a power-of-two boundary of the scale rule
    W[7, :] = 0; W[7, 9] = -7.0 / 1024
    W[8, :] = 0; W[8, 1] = 448.0 * 4 * (1 + 2 ** -20)
    w8, inv = quant_rows_gpu(W)
    torch.cuda.synchronize()
    ref8, ref_inv = F8.quant_rows(W.cpu())
    assert torch.equal(inv.cpu(), ref_inv)
    assert torch.equal(e4m3_values(w8.cpu()), ref8.double())


def test_quant_act_bit_exact():
    x = _randn(1000, 768, seed=2, scale=3.0)
    x[0, :12] = torch.tensor([0.0, 1.0, 1.0625, 1.1875, 55.9, 56.0, 58.0, 1e6, -1e6, 2.0 ** -12, 2.0 ** -13, 7.5 * 2.0 ** -12])   # ties, clamp, subnormals
    got = quant_act_gpu(x, F8.S_LN)
    torch.cuda.synchronize()
    assert torch.equal(e4m3_values(got.cpu()), F8.quant_act(x.cpu(), F8.S_LN).double())


def _operands(M, N, K, seed):
    A, W, bias = _randn(M, K, seed=seed), _randn(N, K, seed=seed + 1, scale=0.05), _randn(N, seed=seed + 2)
    a8 = quant_act_gpu(A, F8.S_LN)
    w8, inv = quant_rows_gpu(W)
    v = (e4m3_values(a8) @ e4m3_values(w8).t()) * (inv.double() / F8.S_LN) + bias.double()      # exact products, fp64 sums
    return a8, w8, inv, bias, v


@pytest.mark.parametrize("M,N,K", [(256, 256, 768), (512, 2304, 768), (256, 768, 3072), (256 * 9, 1024, 1024), (256, 256, 256)])
def test_gemm_fp8_bias(M, N, K):
    lib = _lib.load_fp8()
    a8, w8, inv, bias, v = _operands(M, N, K, 10)
    out = torch.zeros(1, M, N, dtype=torch.float16, device=DEV)
    _chk(lib.vtq_k_gemm_fp8(a8.data_ptr(), K, w8.data_ptr(), inv.data_ptr(), 1.0 / F8.S_LN, M, N, K, 0, bias.data_ptr(), None, None,
                                  out.data_ptr(), M * N, N, 0.0, stream()))
    torch.cuda.synchronize()
    got = planes_value(out)
    assert torch.equal(out[0], v.float().to(torch.float16)) or (out[0].double() - v.float().to(torch.float16).double()).abs().max().item() <= \
        2.0 ** -10 * v.abs().max().item()                                       # the fp16 rounding of the (fp32-accumulated) exact value
    assert (got - v).abs().max().item() < 6e-4 * v.abs().max().item()


def test_gemm_fp8_gelu_e4m3_output():
    lib = _lib.load_fp8()
    M, N, K = 512, 3072, 768
    a8, w8, inv, bias, v = _operands(M, N, K, 20)
    out = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
    _chk(lib.vtq_k_gemm_fp8(a8.data_ptr(), K, w8.data_ptr(), inv.data_ptr(), 1.0 / F8.S_LN, M, N, K, 1, bias.data_ptr(), None, None,
                                  out.data_ptr(), 0, N, F8.S_GELU, stream()))
    torch.cuda.synchronize()
    g = torch.nn.functional.gelu(v) * F8.S_GELU
    got = e4m3_values(out)
    want = F8.to_e4m3(g.float()).double()
    # the kernel rounds its own fp32 value (fp32 sums: ~2e-6 of the tensor's max): one within that noise of a rounding boundary lands on
    # the neighbour -- expected rate ~ noise / grid step ~ 5e-4
    mism = got != want
    print(f"\ne4m3 GELU outputs off the fp64 rounding: {mism.double().mean().item():.2e} of {mism.numel()}")
    assert mism.double().mean().item() < 2e-3
    ulp = torch.maximum(torch.exp2(torch.floor(torch.log2(g.abs().clamp_min(2.0 ** -6))) - 3), torch.tensor(2.0 ** -9, dtype=torch.float64, device=DEV))
    assert ((got - g).abs() <= 0.5 * ulp + 1e-5 * g.abs().max()).all()        # every output IS a correct rounding of a value within fp32 noise


@pytest.mark.parametrize("use_gamma", [False, True])
def test_gemm_fp8_residual(use_gamma):
    lib = _lib.load_fp8()
    M, N, K = 512, 768, 3072
    a8, w8, inv, bias, v = _operands(M, N, K, 30)
    gamma = _randn(N, seed=34) if use_gamma else None
    x0 = _randn(M, N, seed=35)
    x = x0.clone()
    _chk(lib.vtq_k_gemm_fp8(a8.data_ptr(), K, w8.data_ptr(), inv.data_ptr(), 1.0 / F8.S_LN, M, N, K, 2, bias.data_ptr(),
                                  gamma.data_ptr() if use_gamma else None, x.data_ptr(), None, 0, 0, 0.0, stream()))
    torch.cuda.synchronize()
    ref = x0.double() + (gamma.double() * v if use_gamma else v)
    # (the MX MFMA's 128-deep sums are not fp32-exact: ~1.5e-5 of the tensor's max at K = 3072, measured)
    assert (x.double() - ref).abs().max().item() < 5e-5 * ref.abs().max().item()


def build(kw, sd_np, engine_options=0):
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8", engine_options=engine_options)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    return m.to(DEV).eval()


# ---- the engine against the fake-quant oracle ------------------------------------------------------------------------------
# Rounding to a 3-mantissa-bit grid makes the fp8 MODEL discontinuous: an operand that sits within fp32 noise of a rounding
# boundary lands on the other side in one implementation (the MX MFMA's sums are ~1e-5 off the fp64 sum of the same products),
# which moves that operand by a whole grid step (6 %), which flips ~7 % of the roundings of the row it feeds in the next stage...
# measured (tools/fp8_debug.py): 6e-4 of the LayerNorm-1 bytes of layer 0 differ, 17 % of the GELU bytes of the same layer do,
# and by layer 3 the two runs are independent samples of the fp8 model's rounding noise.  An end-to-end 1e-3 gate between ANY
# two implementations of this model is therefore impossible; what is checkable, and checked here, is
#   (1) stage by stage with the engine's OWN stage inputs (teacher forcing): every stage output is the oracle's stage function of
#       those inputs up to boundary flips -- a bounded fraction of bytes, each exactly one grid step off;
#   (2) end to end: the engine's distance to the fp32 model matches the oracle's distance to it, layer by layer (same noise level).


class Probe:
    """vtq_debug_stop_after / vtq_debug_buffers: the engine's workspace after a given stage of a given layer."""

    def __init__(self, model, args, spec, B, N):
        self.m, self.args, self.spec = model, args, spec
        self.H, self.Md = spec.hidden_size, spec.mlp_dim
        self.S, self.nseq = N + spec.num_tokens, 2 * B              # sequences are packed back to back (engine.hip geometry())
        self.lib, self.hip = _lib.load_fp8(), C.CDLL("libamdhip64.so")
        with torch.no_grad():
            model(*args)                                            # creates the engine

    def grab(self, layer, stage):
        _chk(self.lib.vtq_debug_stop_after(self.m._engine, layer * 7 + stage))
        with torch.no_grad():
            self.m(*self.args)
        torch.cuda.synchronize()
        _chk(self.lib.vtq_debug_stop_after(self.m._engine, -1))
        x, ln, big, rows = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _chk(self.lib.vtq_debug_buffers(self.m._engine, C.byref(x), C.byref(ln), C.byref(big), C.byref(rows)))
        R, H = rows.value, self.H

        def copy(ptr, nbytes):
            t = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
            assert self.hip.hipMemcpy(C.c_void_p(t.data_ptr()), ptr, C.c_size_t(nbytes), 3) == 0
            return t.cpu()
        return dict(x=copy(x, R * H * 4).view(torch.float32).view(R, H), ln=copy(ln, R * H * 2), big=copy(big, R * max(3 * H, self.Md) * 2), R=R)

    def seqs(self, t):
        return t[:self.nseq * self.S].view(self.nseq, self.S, -1)

    def e4(self, raw, width, R):
        return self.seqs(raw[:R * width].view(torch.float8_e4m3fn).float().view(R, width))


def grid_step(v):
    """e4m3 grid spacing at |v| (v already scaled)."""
    return torch.maximum(torch.exp2(torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -6))) - 3), torch.tensor(2.0 ** -9))


def check_bytes(name, got, want_unrounded, max_flips, noise=2e-5):
    """got: the engine's e4m3 values; want_unrounded: the oracle's scaled value before rounding.  A byte may sit on the neighbouring
    grid point (the two sides' inputs differ by arithmetic noise), never further -- where "further" allows for that noise itself,
    `noise` x the tensor's max: with calibrated scales the tensor's max maps to 112 .. 224, so the fp16 noise of the attention
    (3e-4 of the max) is many steps of the SUBNORMAL grid (2^-9) for elements near zero, while still far below a grid step of any
    element that matters."""
    want = F8.to_e4m3(want_unrounded)
    flips = (got != want).float().mean().item()
    slack = noise * want_unrounded.abs().max()
    worst = (((got - want).abs() - slack).clamp_min(0) / grid_step(want_unrounded)).max().item()
    print(f"   {name:12s} bytes off the oracle's rounding: {flips:.2e}   worst {worst:.2f} grid steps beyond the noise allowance")
    assert flips < max_flips, (name, flips)
    assert worst <= 1.0 + 1e-6, (name, worst)               # never more than the neighbouring grid point


def check_f32(name, got, want, tol):
    d = ((got - want).abs().max() / want.abs().max()).item()
    print(f"   {name:12s} max error / max: {d:.2e}")
    assert d < tol, (name, d)


@pytest.mark.parametrize("name", ["c1_b2_n50", "scales3_b2_n40", "vitl_b2_n70", "refdefault_b2_n64"])
def test_fp8_stages_teacher_forced(name):
    import math
    import torch.nn.functional as Fn
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    pr = Probe(model, (p, ps, sc), spec, int(g["B"]), int(g["N"]))
    s8 = F8.Scales.from_engine(model.fp8_scales())          # calibrated on this batch by the first forward (engine.hip fp8_stage)
    sdt = O.to_torch(sd)
    H, Md, nh = spec.hidden_size, spec.mlp_dim, spec.num_heads
    dh = H // nh
    for layer in sorted({0, spec.num_layers // 2, spec.num_layers - 1}):
        S_LN1, S_ATT, S_LN2, S_GELU = s8.ln1[layer], s8.att[layer], s8.ln2[layer], s8.gelu[layer]
        print(f"\n[{name}] layer {layer}   scales: LN1 x{S_LN1:g}  attention x{S_ATT:g}  LN2 x{S_LN2:g}  GELU x{S_GELU:g}")
        pre = f"transformer.encoder.layers.{layer}."
        st = [pr.grab(layer, k) for k in range(7)]
        R = st[0]["R"]
        x0 = pr.seqs(st[0]["x"])                                                    # LayerNorm 1 leaves the stream untouched
        ln1 = pr.e4(st[0]["ln"], H, R)
        check_bytes("LayerNorm 1", ln1, O._layer_norm(x0, sdt[pre + "attention_norm.weight"], sdt[pre + "attention_norm.bias"]) * S_LN1, 3e-3)
        qkv = pr.seqs(st[1]["big"][:R * 3 * H * 2].view(torch.float16).float().view(R, 3 * H))
        want = torch.cat([F8.linear8(ln1.double(), S_LN1, sdt[f"{pre}attn.{n}.weight"].double(), sdt[f"{pre}attn.{n}.bias"].double())
                          for n in ("query", "key", "value")], -1)
        check_f32("QKV (fp16)", qkv.double(), want, 1e-3)
        q, k, v = (t.double().view(pr.nseq, pr.S, nh, dh).permute(0, 2, 1, 3) for t in qkv.split(H, -1))
        ctx = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v).permute(0, 2, 1, 3).reshape(pr.nseq, pr.S, H)
        ctx8 = pr.e4(st[2]["ln"], H, R)
        check_bytes("attention", ctx8, ctx.float() * S_ATT, 5e-2, noise=1e-3)                # single-fp16 P and V: 3e-4 of noise against a 6 % grid
        h = F8.linear8(ctx8.double(), S_ATT, sdt[pre + "attn.out.weight"].double(), sdt[pre + "attn.out.bias"].double())
        if spec.use_layer_scale:
            h = h * sdt[pre + "ls1.gamma"].double()
        x1 = pr.seqs(st[3]["x"])
        check_f32("x + attn", x1.double(), x0.double() + h, 1e-4)
        ln2 = pr.e4(st[4]["ln"], H, R)
        check_bytes("LayerNorm 2", ln2, O._layer_norm(x1, sdt[pre + "ffn_norm.weight"], sdt[pre + "ffn_norm.bias"]) * S_LN2, 3e-3)
        pre_act = F8.linear8(ln2.double(), S_LN2, sdt[pre + "ffn.fc1.weight"].double(), sdt[pre + "ffn.fc1.bias"].double())
        g8 = pr.e4(st[5]["big"], Md, R)
        check_bytes("fc1 + GELU", g8, (Fn.gelu(pre_act) * S_GELU).float(), 3e-3, noise=5e-5)
        h = F8.linear8(g8.double(), S_GELU, sdt[pre + "ffn.fc2.weight"].double(), sdt[pre + "ffn.fc2.bias"].double())
        if spec.use_layer_scale:
            h = h * sdt[pre + "ls2.gamma"].double()
        check_f32("x + mlp", pr.seqs(st[6]["x"]).double(), x1.double() + h, 1e-4)


def test_fp8_patch_embedding_teacher_forced():
    """Embeddings: the stream entering layer 0 against the oracle's (no rounding of computed values upstream: only the inputs)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("scales3_b2_n40")
    model = build(kw, sd)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    pr = Probe(model, (p, ps, sc), spec, int(g["B"]), int(g["N"]))
    x = pr.seqs(pr.grab(0, 0)["x"])
    sdt = O.to_torch(sd)
    pc, psc, scc = split_inputs(patches, pos, scales)
    s8 = F8.Scales.from_engine(model.fp8_scales())
    want = torch.cat([F8.embeddings(sdt, spec, pc[i], psc[i], scc[i], s8) for i in range(2)])
    check_f32("embeddings", x, want, 1e-4)


FP8_CASES = [c for c in E2E_CASES if c not in ("adapters_b2_n40", "preemb_b3_n60")]          # adapters are rejected in the fp8 mode (test below)


def test_fp8_rejects_adapters():
    with pytest.raises(NotImplementedError):
        VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, num_adapters=1, pretrained=False), precision="fp8")


@pytest.mark.parametrize("name", FP8_CASES)
def test_fp8_end_to_end_noise_level(name):
    """The engine's scores and per-layer CLS rows are as far from the fp32 model as the oracle's fp8 model is (same rounding-noise
    level); the distances themselves are printed, not gated: this mode makes no 1e-3 claim."""
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    B, L, T, H = int(g["B"]), spec.num_layers, spec.num_tokens, spec.hidden_size
    trace = torch.zeros(L + 1, 2 * B, T, H, device=DEV)
    with torch.no_grad():
        q, aux = model(p, ps, sc, _trace=trace)
    q = q.cpu().numpy()
    assert aux is None and q.shape == (B,) and np.isfinite(q).all()
    pc, psc, scc = split_inputs(patches, pos, scales)
    tr8, tr32 = {}, {}
    s8 = F8.Scales.from_engine(model.fp8_scales())          # the engine's calibrated scales (first forward = this batch)
    q8 = F8.vtamiq_forward(O.to_torch(sd), spec, pc, psc, scc, trace=tr8, s8=s8)[0].numpy()
    O.vtamiq_forward(O.to_torch(sd), spec, pc, psc, scc, trace=tr32)
    t32 = torch.cat([tr32["tokens_ref"], tr32["tokens_dist"]], dim=1).numpy()
    got, t8 = trace.cpu().numpy(), tr8["tokens"].numpy()
    dev = lambda a, l: float(np.abs(a[l] - t32[l]).max() / np.abs(t32[l]).max())
    print(f"\n[{name} fp8] scores: engine vs fp32 {gate_error(q, g['q']):.3e}   oracle-fp8 vs fp32 {gate_error(q8, g['q']):.3e}   "
          f"engine vs oracle-fp8 {gate_error(q, q8):.3e}   (reported)")
    for l in range(1, L + 1):
        dg, do = dev(got, l), dev(t8, l)
        if l in (1, L // 2, L):
            print(f"   CLS rows after layer {l}: engine vs fp32 {dg:.3e}   oracle-fp8 vs fp32 {do:.3e}")
        assert dg < 2.0 * do + 2e-3, (l, dg, do)
    assert np.abs(got[0] - t8[0]).max() <= 1e-6 * np.abs(t8[0]).max()


def _flat(s8):
    return [s8.patch] + [v for i in range(len(s8.ln1)) for v in (s8.ln1[i], s8.att[i], s8.ln2[i], s8.gelu[i])]


@pytest.mark.parametrize("name", ["c1_b2_n50", "scales3_b2_n40"])
def test_fp8_calibration_matches_the_oracles(name):
    """The scales the engine calibrates on its first batch against oracle.fp8_oracle.calibrate on the same batch: the same power
    of two at (nearly) every quantisation point -- the maxima the two see differ by rounding noise, so a point whose maximum sits
    on a power-of-two boundary of the rule may land one binade apart, never more."""
    import math
    g, kw, spec, sd, (patches, pos, scales) = load_case(name)
    model = build(kw, sd)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q = model(p, ps, sc)[0]
    model.check_inputs()                                      # nothing clamped with calibrated scales
    got = _flat(F8.Scales.from_engine(model.fp8_scales()))
    want = _flat(F8.calibrate(O.to_torch(sd), spec, *split_inputs(patches, pos, scales)))
    assert len(got) == len(want) == 1 + 4 * spec.num_layers
    ratio = [abs(math.log2(a / b)) for a, b in zip(got, want)]
    print(f"\n[{name}] engine scales {got}\n oracle scales {want}")
    assert max(ratio) <= 1.0 and sum(r == 0 for r in ratio) >= 0.85 * len(ratio)
    assert all(math.frexp(v)[0] == 0.5 for v in got)          # powers of two
    # explicit scales round-trip through the ABI, and a second forward is bit-identical (the scales are model state, not batch state)
    model.set_fp8_scales(model.fp8_scales())
    with torch.no_grad():
        assert torch.equal(model(p, ps, sc)[0], q)


def test_fp8_static_scales_saturate_on_trained_like_weights_calibrated_ones_do_not():
    """tests.helpers.stress_state (LayerNorm gains x8 on outlier channels, fc2 bias +2): the round-2 constants (LayerNorm x8, GELU x4)
    push values past e4m3's 448 -- clamped, and now REPORTED (error word bit 2) -- while scales calibrated on the batch do not."""
    from tests.helpers import stress_state
    from vtamiq_amd import synth
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=4, pretrained=False))
    spec = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8").spec
    sd = stress_state(spec, 9, qk=5.0, outlier=64.0)
    patches, pos, scales = synth.make_inputs(spec, 2, 80, 21)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    m_static = build(kw, sd, engine_options=_lib.OPT_FP8_STATIC_SCALES)
    with torch.no_grad():
        q_static = m_static(p, ps, sc)[0]
    with pytest.raises(FloatingPointError, match="clamped"):
        m_static.check_inputs()
    m_cal = build(kw, sd)
    with torch.no_grad():
        q_cal = m_cal(p, ps, sc)[0]
    m_cal.check_inputs()                                      # no flag
    s8 = m_cal.fp8_scales()
    assert min(s8["ln1"] + s8["ln2"]) < 8.0                   # calibration had to lower a LayerNorm scale below the constant
    q32 = O.vtamiq_forward(O.to_torch(sd), spec, *split_inputs(patches, pos, scales))[0].numpy()
    e_static, e_cal = gate_error(q_static.cpu().numpy(), q32), gate_error(q_cal.cpu().numpy(), q32)
    print(f"\nstressed weights, fp8 vs fp32 oracle: static scales {e_static:.3e} (saturated), calibrated {e_cal:.3e}")
    assert np.isfinite(q_cal.cpu().numpy()).all()


def test_fp8_batch_invariance_and_order():
    """Each pair's score is independent of its neighbours in the batch (the scales are calibrated once, on the first batch, and are
    model state from then on: nothing is batch-dependent)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c2shape_b4_n500")
    model = build(kw, sd)
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    with torch.no_grad():
        q = model(p, ps, sc)[0].cpu().numpy()
        perm = [2, 0, 3, 1]
        qp = model(tuple(t[perm] for t in p), tuple(t[perm] for t in ps), tuple(None if t is None else t[perm] for t in sc))[0].cpu().numpy()
        q1 = model(tuple(t[1:2] for t in p), tuple(t[1:2] for t in ps), tuple(None if t is None else t[1:2] for t in sc))[0].cpu().numpy()
    assert np.array_equal(qp, q[perm])
    assert np.array_equal(q1, q[1:2])


def test_fp8_attention_kernels_agree():
    """The e4m3 context bytes (and the activation-range report behind the calibration) of the pipelined attention kernel equal the
    4-wave kernel's: same scores bit for bit with either forced (the library's own rule keeps single-plane attention on the 4-wave one)."""
    from vtamiq_amd import _lib
    lib = _lib.load_fp8()
    g, kw, spec, sd, (patches, pos, scales) = load_case("c2shape_b4_n500")
    p, ps, sc = split_inputs(patches, pos, scales, device=DEV)
    got = []
    for variant in (0, 1):
        lib.vtq_debug_attention_variant(variant)
        try:
            model = build(kw, sd)                       # calibrates on its first forward, through the forced kernel
            with torch.no_grad():
                got.append((model(p, ps, sc)[0].cpu().numpy(), model.fp8_scales()))
                model.check_inputs()                    # no clamped activation with either kernel
        finally:
            lib.vtq_debug_attention_variant(-1)
    assert np.array_equal(got[0][0], got[1][0])
    assert got[0][1] == got[1][1]


def test_fp8_pairwise_triplets_match_two_calls():
    """SURVEY 8f-2 in the fp8 mode: the fused (ref, dist1, dist2) entry point reproduces the two model calls bit for bit."""
    from vtamiq_amd import synth
    kw = dict(vit_config=dict(variant="ViT-B16", num_keep_layers=3, num_scales=3))
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8")
    sd = synth.make_state_dict(m.spec, 31)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    B, N = 4, 60
    patches, pos, scales = synth.make_inputs(m.spec, B, N, 32, aligned=False)
    rs = np.random.RandomState(3)
    d2 = np.clip(patches[:, 0] + 0.2 * rs.randn(*patches[:, 0].shape), -1, 1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    pr, pd1, pd2 = t(patches[:, 0]), t(patches[:, 1]), t(d2)
    qr, qd1 = t(pos[:, 0]), t(pos[:, 1])
    sr, sd1 = t(scales[:, 0]).float(), t(scales[:, 1]).float()
    with torch.no_grad():
        q1 = m((pr, pd1), (qr, qd1), (sr, sd1))[0]
        q2 = m((pr, pd2), (qr, qd1), (sr, sd1))[0]
        f1, f2 = m.forward_pairwise((pr, pd1, pd2), (qr, qd1, qd1), (sr, sd1, sd1))
    assert torch.equal(f1, q1) and torch.equal(f2, q2) and bool(torch.isfinite(f1).all())


def test_fp8_repeated_forwards_are_bitwise_identical():
    """(moved here from test_gpu_parity.py with the mode) the same batch 25 times, interleaved with another batch: identical bits."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c2shape_b4_n500")
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to("cuda").eval()
    p, ps, sc = split_inputs(patches, pos, scales, device="cuda")
    other = tuple(t.flip(0).contiguous() for t in p)
    with torch.no_grad():
        q0 = m(p, ps, sc)[0].clone()
        for i in range(25):
            if i % 3 == 0:
                m(other, ps, sc)
            assert torch.equal(m(p, ps, sc)[0], q0), i


def test_fp8_auto_scales_never_outlive_the_weights_they_fit():
    """auto-calibrated scales belong to one (engine, weights) pair.  forward -> .to() (same device: weights re-packed) ->
    forward -> load_state_dict of 8x larger weights -> forward: the scales CHANGE (they were fitted to the old activations), the saved
    copy follows, and no stale set is ever re-installed; user-installed scales survive the same sequence."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    p, ps, sc = split_inputs(patches, pos, scales, device="cuda")
    state = {k: torch.from_numpy(v) for k, v in sd.items()}
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8")
    m.load_state_dict(state)
    m = m.to("cuda").eval()
    with torch.no_grad():
        m(p, ps, sc)
        s0 = m.fp8_scales()
        m = m.to("cuda")                                   # re-pack: the engine recalibrates on the next forward, the saved copy is dropped
        assert m.__dict__.get("_fp8_saved") is None or m._weights_sig is None
        m(p, ps, sc)
        assert m.fp8_scales() == s0                        # same weights, same batch: the same scales again
        big = {k: (v * 8.0 if k.endswith("ffn.fc1.weight") or k.endswith("attention_norm.weight") else v) for k, v in state.items()}
        m.load_state_dict(big)
        m(p, ps, sc)
        s1 = m.fp8_scales()
        assert s1 != s0 and m.__dict__["_fp8_saved"] == s1
        assert all(a <= b for a, b in zip(s1["ln1"], s0["ln1"])) and s1["ln1"] != s0["ln1"]      # larger activations: smaller scales
        m.set_fp8_scales(s0)                               # the user's scales: kept across a reload
        m.load_state_dict(state)
        m(p, ps, sc)
        assert m.fp8_scales() == s0


def test_fp8_validate_inputs_raises_after_every_forward():
    """precision fp8 with validate_inputs keeps the documented contract -- IndexError for a position outside [0, 1)."""
    g, kw, spec, sd, (patches, pos, scales) = load_case("c1_b2_n50")
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision="fp8")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to("cuda").eval()
    m.validate_inputs = True
    p, ps, sc = split_inputs(patches, pos, scales, device="cuda")
    with torch.no_grad():
        m(p, ps, sc)
        bad = (ps[0].clone(), ps[1].clone())
        bad[1][0, 3, 0] = 1.5
        with pytest.raises(IndexError):
            m(p, bad, sc)
        m(p, ps, sc)                                       # the flag was consumed: a clean batch passes again

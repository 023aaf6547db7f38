"""Per-kernel parity on the GPU, through the C ABI, against fp64 torch references of the same op."""

import pytest
import torch

from vtamiq_amd import _lib
from tests.gpu_util import FORMATS, elt_dtype, num_code, planes_of, planes_value, stream, to_planes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _randn(*s, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


FMTS = ["bf16", "bf16x3", "fp16", "fp16x2", "fp16x3"]
# tile shapes of a GEMM launch (include/vtamiq_hip.h vtq_debug_gemm_variant): the library's rule, the persistent 256x256 kernel, and the
# small tiles of csrc/gemm_st.hip
TILE_RULE, TILE_256, TILE_ST = -1, 0, (1, 2, 3)


@pytest.fixture(params=[TILE_RULE, TILE_256], ids=["tile_rule", "tile_256"])
def tile(request):
    """Run a GEMM test twice: with the shape the library's rule picks for it (small-tile kernels for these small M) and with the
    persistent 256x256 kernel forced."""
    lib = _lib.load()
    _lib.check(lib.vtq_debug_gemm_variant(request.param))
    yield request.param
    _lib.check(lib.vtq_debug_gemm_variant(TILE_RULE))
# error of a 16-bit OUTPUT relative to the tensor's max: one plane rounds to 8 / 11 bits, two planes carry 16 / 22
OUT_TOL = {"bf16": 1e-2, "fp16": 2e-3, "bf16x3": 1e-4, "fp16x2": 2e-5, "fp16x3": 2e-5}


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("role", ["a", "w"])
def test_split(fmt, role):
    x = _randn(1000, 768, seed=1)
    x[0, :8] = torch.tensor([1e-6, -3e-7, 6.1e-5, 0.0, 65000.0 if fmt.startswith("fp16") else 1e30, -1.0, 2.0 ** -24, 1e-8])   # subnormals, range
    p = to_planes(x, fmt, role)
    dt = elt_dtype(fmt)
    hi = x.to(dt)
    assert p.shape[0] == planes_of(fmt, role) and torch.equal(p[0], hi)
    if p.shape[0] == 2:
        assert torch.equal(p[1], (x - hi.float()).to(dt))
        rel = 2e-5 if dt == torch.bfloat16 else 3e-7
        assert ((planes_value(p) - x.double()).abs() <= rel * x.abs().double() + 1e-7).all()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("M,N,K", [(256, 256, 768), (512, 768, 768), (768, 2304, 768), (256, 768, 3072), (512, 1024, 1024), (2048, 1024, 256)])
def test_gemm_bias(fmt, M, N, K, tile):
    lib = _lib.load()
    A, W, bias = _randn(M, K, seed=2), _randn(N, K, seed=3, scale=0.05), _randn(N, seed=4)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 0, bias.data_ptr(), None, None,
                              out.data_ptr(), M * N, N, stream()))
    torch.cuda.synchronize()
    ref = planes_value(Ap) @ planes_value(Wp).t() + bias.double()       # the operands as the kernel sees them
    tol = OUT_TOL[fmt]
    got = planes_value(out)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    # asymmetric data + non-square shapes: a transposed or permuted tile would be O(1) wrong
    assert torch.allclose(got, ref, rtol=0, atol=tol * ref.abs().max().item())
    if FORMATS[fmt][1] == 3:     # the split planes reproduce the fp32 operands: also close to the exact product
        exact = A.double() @ W.double().t() + bias.double()
        assert (got - exact).abs().max().item() < (1e-4 if fmt == "bf16x3" else 2e-5) * exact.abs().max().item()


@pytest.mark.parametrize("fmt", FMTS)
def test_gemm_many_tiles_per_workgroup(fmt):
    """More than 256 tiles: every workgroup of the persistent launch walks several tiles (DMA ring chained across tile
    boundaries, half tiles closing the lists); checked on sampled rows against fp64."""
    lib = _lib.load()
    M, N, K = 256 * 70, 2304, 768
    A, W, bias = _randn(M, K, seed=31), _randn(N, K, seed=32, scale=0.05), _randn(N, seed=33)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=DEV)
    for _ in range(2):
        _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 0, bias.data_ptr(), None, None,
                                  out.data_ptr(), M * N, N, stream()))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, M, 97), torch.tensor([127, 128, 255, 256, M - 129, M - 128, M - 1])]).to(DEV)
    ref = planes_value(Ap)[rows] @ planes_value(Wp).t() + bias.double()
    got = planes_value(out)[rows]
    assert (got - ref).abs().max().item() < OUT_TOL[fmt] * ref.abs().max().item()


@pytest.mark.parametrize("fmt", FMTS)
def test_gemm_gelu(fmt, tile):
    lib = _lib.load()
    M, N, K = 256, 3072, 768
    A, W, bias = _randn(M, K, seed=5), _randn(N, K, seed=6, scale=0.05), _randn(N, seed=7)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 1, bias.data_ptr(), None, None,
                              out.data_ptr(), M * N, N, stream()))
    torch.cuda.synchronize()
    pre = (planes_value(Ap) @ planes_value(Wp).t() + bias.double())
    ref = torch.nn.functional.gelu(pre)
    got = planes_value(out)
    assert (got - ref).abs().max().item() < OUT_TOL[fmt] * ref.abs().max().item()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("use_gamma", [False, True])
@pytest.mark.parametrize("M", [512, 256 * 41])
def test_gemm_residual(fmt, use_gamma, M, tile):
    lib = _lib.load()
    N, K = 768, 3072 if M == 512 else 768
    A, W, bias = _randn(M, K, seed=8), _randn(N, K, seed=9, scale=0.02), _randn(N, seed=10)
    gamma = _randn(N, seed=11) if use_gamma else None
    x0 = _randn(M, N, seed=12)
    x = x0.clone()
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 2, bias.data_ptr(),
                              gamma.data_ptr() if use_gamma else None, x.data_ptr(), None, 0, 0, stream()))
    torch.cuda.synchronize()
    rows = torch.arange(0, M, 1 if M == 512 else 61, device=DEV)
    h = planes_value(Ap)[rows] @ planes_value(Wp).t() + bias.double()
    ref = x0[rows].double() + (gamma.double() * h if use_gamma else h)
    assert (x[rows].double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (1024, 768, 3072), (1280, 2304, 768), (512, 1024, 256)])
def test_gemm_tile_shapes_agree_bitwise(fmt, epi, M, N, K):
    """The bitwise contract of csrc/gemm_st.hip: every tile shape gives every output element the MFMA sequence and the epilogue
    arithmetic of the 256x256 kernel, so the outputs are IDENTICAL -- which is what lets vtq_k_gemm_tile_rule choose by (M, N) without
    the scores depending on the batch size."""
    lib = _lib.load()
    A, W, bias, gamma = _randn(M, K, seed=41), _randn(N, K, seed=42, scale=0.05), _randn(N, seed=43), _randn(N, seed=44)
    x0 = _randn(M, N, seed=45)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    outs = {}
    try:
        for v in (TILE_256,) + TILE_ST:
            _lib.check(lib.vtq_debug_gemm_variant(v))
            out = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=DEV)
            x = x0.clone()
            for _ in range(2 if epi != 2 else 1):          # a second launch over the same output: same bits again
                _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), epi, bias.data_ptr(),
                                          gamma.data_ptr() if epi == 2 else None, x.data_ptr() if epi == 2 else None,
                                          out.data_ptr() if epi != 2 else None, M * N, N, stream()))
            torch.cuda.synchronize()
            outs[v] = x if epi == 2 else out
    finally:
        _lib.check(lib.vtq_debug_gemm_variant(TILE_RULE))
    base = outs[TILE_256]
    assert bool(torch.isfinite(base.float()).all()) and float(base.float().abs().max()) > 0
    for v in TILE_ST:
        assert torch.equal(outs[v].view(torch.int32 if epi == 2 else torch.int16), base.view(torch.int32 if epi == 2 else torch.int16)), (v, fmt, epi)


def test_gemm_tile_rule():
    """Host-only rule: small tiles only while the 256x256 form would leave most CUs idle; never for fp8 operands."""
    lib = _lib.load()
    rule = lambda M, N, K, fmt="fp16x3": lib.vtq_k_gemm_tile_rule(M, N, K, _lib.NUM[fmt])
    assert rule(1024, 768, 3072) == 1 and rule(2048, 768, 768) == 2 and rule(4096, 768, 3072) == 3        # B = 1, 2, 4 pairs: out-proj / fc2
    assert rule(1024, 2304, 768) == 3 and rule(1024, 3072, 768) == 3                                       # B = 1: QKV, fc1
    assert rule(2048, 2304, 768) == 0 and rule(8192, 768, 768) == 0 and rule(32256, 768, 768) == 0         # 64 tiles of 256x256 and up
    assert rule(1024, 768, 768, "fp8") == 0 and rule(1024, 768, 768, "bf16") == 1
    assert lib.vtq_k_gemm_tile_rule(1024, 768, 768, 99) == -1


@pytest.mark.parametrize("fmt", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("M,K", [(128, 768), (256 * 5, 768), (128 * 3, 3072), (128 * 259, 768), (128 * 257, 128), (128 * 2, 96 + 32)])
@pytest.mark.parametrize("use_gamma,use_ln", [(True, True), (False, True), (True, False)])
def test_gemm_rowln_is_the_two_launch_path_bit_for_bit(fmt, M, K, use_gamma, use_ln):
    """The whole-row residual GEMM with LayerNorm in its epilogue (gemm_rowln.hip) against the path it replaces -- vtq_k_gemm epilogue 2
    (x += gamma * (A W^T + bias)) followed by vtq_k_layernorm -- BITWISE: the accumulation order per element and the LayerNorm arithmetic
    are the same by construction.  Shapes: one tile, several tiles, K = 3072 (fc2), more tiles than CUs (a workgroup walks two), the
    smallest K the kernel takes (4 K tiles) -- and the two-launch path is itself checked against fp64 by test_gemm_residual / test_layernorm."""
    lib = _lib.load()
    N = 768
    Mp = (M + 255) // 256 * 256                                # the 256 x 256 kernel needs M % 256 == 0; the row kernel M % 128 == 0
    A, W, bias = _randn(Mp, K, seed=21), _randn(N, K, seed=22, scale=0.03), _randn(N, seed=23)
    gamma = _randn(N, seed=24) + 1.0 if use_gamma else None
    lw, lb = _randn(N, seed=25) + 1.0, _randn(N, seed=26)
    x0 = _randn(Mp, N, seed=27, scale=2.0)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    x_ref = x0.clone()
    _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, Mp, N, K, num_code(fmt), 2, bias.data_ptr(),
                              gamma.data_ptr() if use_gamma else None, x_ref.data_ptr(), None, 0, 0, stream()))
    out_ref = torch.zeros((2, Mp, N), dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_layernorm(x_ref.data_ptr(), lw.data_ptr(), lb.data_ptr(), out_ref.data_ptr(), Mp * N, Mp, N, FORMATS[fmt][0], 2, stream()))
    x = x0.clone()
    out = torch.full((2, Mp, N), 7.0, dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), Mp * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(),
                                    gamma.data_ptr() if use_gamma else None, x.data_ptr(), lw.data_ptr() if use_ln else None,
                                    lb.data_ptr() if use_ln else None, out.data_ptr() if use_ln else None, Mp * N, stream()))
    torch.cuda.synchronize()
    assert torch.equal(x[:M], x_ref[:M])
    assert torch.equal(x[M:], x0[M:])                          # rows beyond M untouched
    if use_ln:
        assert torch.equal(out[:, :M].view(torch.int16), out_ref[:, :M].view(torch.int16))
    assert bool((out[:, M:] == 7.0).all())
    if not use_ln:
        assert bool((out == 7.0).all())


def test_gemm_rowln_repeats_bit_for_bit():
    """40 launches on the same operands (four waves with private DMA rings, counted waits, LDS reuse between the main loop and the
    epilogue images, two tiles per workgroup: a race would show as a differing bit)."""
    lib = _lib.load()
    fmt, M, K, N = "fp16x3", 128 * 300, 768, 768
    A, W, bias = _randn(M, K, seed=31), _randn(N, K, seed=32, scale=0.03), _randn(N, seed=33)
    lw, lb, x0 = _randn(N, seed=34) + 1.0, _randn(N, seed=35), _randn(M, N, seed=36)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    first = None
    for it in range(40):
        x = x0.clone()
        out = torch.zeros((2, M, N), dtype=torch.float16, device=DEV)
        _lib.check(lib.vtq_k_gemm_rowln(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, K, num_code(fmt), bias.data_ptr(), None, x.data_ptr(),
                                        lw.data_ptr(), lb.data_ptr(), out.data_ptr(), M * N, stream()))
        torch.cuda.synchronize()
        if first is None:
            first = (x, out)
        else:
            assert torch.equal(x, first[0]) and torch.equal(out.view(torch.int16), first[1].view(torch.int16)), it


@pytest.mark.parametrize("fmt", ["bf16", "bf16x3", "fp16", "fp16x3"])
@pytest.mark.parametrize("H", [768, 1024])
def test_layernorm(fmt, H):
    lib = _lib.load()
    rows = 515
    x = _randn(rows, H, seed=13, scale=3.0) + 0.7
    w, b = _randn(H, seed=14) + 1.0, _randn(H, seed=15)
    npl = planes_of(fmt, "a")
    out = torch.zeros((npl, rows, H), dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), rows * H, rows, H, FORMATS[fmt][0], npl,
                                   stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (H,), w.double(), b.double(), 1e-6)
    got = planes_value(out)
    tol = {"bf16": 8e-3, "fp16": 1e-3, "bf16x3": 3e-5, "fp16x3": 3e-6}[fmt]
    assert (got - ref).abs().max().item() < tol * ref.abs().max().item()


def _attention_ref(qkv, nseq, S, S_pad, H):
    nh = H // 64
    x = qkv.view(nseq, S_pad, 3, nh, 64)[:, :S]
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(nseq, S, H)


@pytest.mark.parametrize("fmt", ["bf16", "bf16x3", "fp16", "fp16x3"])
@pytest.mark.parametrize("nseq,S,H", [(4, 501, 768), (3, 51, 768), (2, 1025, 1024), (2, 64, 768), (2, 509, 768), (5, 521, 768), (3, 9, 768),
                                      (2, 96, 768)])
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention(fmt, nseq, S, H, packed, variant):
    """variant 0 = the 4-wave kernel, 1 = the 8-wave software-pipelined kernel, 2 = split: the pipelined kernel on the full 256-row query
    blocks and the 4-wave kernel on the rows behind them (forced; sequences shorter than 256 or a multiple of it run form 1; the
    library's own rule picks per shape)."""
    lib = _lib.load()
    lib.vtq_debug_attention_variant(variant)
    try:
        _attention_case(lib, fmt, nseq, S, H, packed)
    finally:
        lib.vtq_debug_attention_variant(-1)


def _attention_case(lib, fmt, nseq, S, H, packed, spike_row=None):
    # sequence pitch: the engine packs sequences back to back (pitch = S: the last key tile / query block of a sequence runs
    # into the next one and is masked / not stored); a padded pitch must work as well
    S_pad = S if packed else (S + 31) // 32 * 32
    rows = nseq * S_pad + 128
    qkv = _randn(rows, 3 * H, seed=16, scale=1.5)
    # a spike so that the running max moves late in the sequence (online-softmax rescale path)
    qkv[S - 3 if spike_row is None else spike_row, H:H + 64] *= 6.0
    P = to_planes(qkv, fmt, "a")
    npl = P.shape[0]
    out = torch.zeros((npl, rows, H), dtype=elt_dtype(fmt), device=DEV)
    _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S_pad, H, num_code(fmt), stream()))
    torch.cuda.synchronize()
    ref = _attention_ref(planes_value(P)[: nseq * S_pad], nseq, S, S_pad, H)
    got = planes_value(out)[: nseq * S_pad].view(nseq, S_pad, H)[:, :S]
    tol = {"bf16": 1.5e-2, "fp16": 3e-3, "bf16x3": 2e-4, "fp16x3": 1e-5}[fmt]
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    return out


@pytest.mark.parametrize("fmt", ["fp16x3", "fp16", "bf16x3"])
@pytest.mark.parametrize("nseq,S", [(24, 501), (5, 9), (6, 64), (4, 521), (3, 130)])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_is_not_reached_by_nan_rows_of_the_next_sequence(fmt, nseq, S, variant):
    """Sequences are packed back to back, so the masked keys of a sequence's last key tile are the first rows of the NEXT sequence.  Their
    probabilities are 0, but 0 x NaN = NaN: the kernels zero the V rows of masked keys in the tile's LDS image.  With every row of sequence
    k (and the slack rows behind the last sequence) set to NaN / inf, the outputs of all OTHER sequences are the same bits as without."""
    lib = _lib.load()
    H = 768
    rows = nseq * S + 128
    qkv = _randn(rows, 3 * H, seed=41, scale=1.5)
    P = to_planes(qkv, fmt, "a")
    npl = P.shape[0]
    lib.vtq_debug_attention_variant(variant)
    try:
        clean = torch.zeros((npl, rows, H), dtype=elt_dtype(fmt), device=DEV)
        _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, clean.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
        k = nseq // 2
        Pb = P.clone()
        Pb[:, k * S:(k + 1) * S] = float("nan")
        Pb[:, nseq * S:] = float("inf")                       # the rows behind the last sequence (its last tile reads them)
        dirty = torch.zeros_like(clean)
        _lib.check(lib.vtq_k_attention(Pb.data_ptr(), rows * 3 * H, dirty.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
        torch.cuda.synchronize()
    finally:
        lib.vtq_debug_attention_variant(-1)
    for s_ in range(nseq):
        a, b = clean[:, s_ * S:(s_ + 1) * S].view(torch.int16), dirty[:, s_ * S:(s_ + 1) * S].view(torch.int16)
        if s_ == k:
            assert bool(torch.isnan(dirty[:, s_ * S:(s_ + 1) * S].float()).all())
        else:
            assert torch.equal(a, b), s_
            assert bool(torch.isfinite(dirty[:, s_ * S:(s_ + 1) * S].float()).all())


@pytest.mark.parametrize("fmt", ["fp16x3", "fp16", "bf16x3"])
@pytest.mark.parametrize("nseq,S,H", [(24, 501, 768), (6, 1025, 1024), (3, 257, 768), (2, 2501, 768), (7, 64, 768), (32, 521, 768), (5, 575, 768)])
def test_attention_kernels_agree_bitwise(fmt, nseq, S, H):
    """The two kernels -- and the split form that gives the rows behind the last full 256-row block to the 4-wave kernel (521: the
    reference-default topology) -- run the same arithmetic in the same order per query row: identical bits, also with the row maximum
    moving late and sitting in either half-wave (keys whose score lands in lanes 32..63 exposed a dropped v_permlane32_swap in round 3),
    on ragged last blocks (257, 1025: one valid row in the last 256-row block) and across the block seams of a persistent workgroup."""
    lib = _lib.load()
    for spike in (S - 3, S - 7 if S > 7 else 0):
        outs = []
        for variant in (0, 1, 2):
            lib.vtq_debug_attention_variant(variant)
            try:
                outs.append(_attention_case(lib, fmt, nseq, S, H, True, spike_row=spike).view(torch.int16).clone())
            finally:
                lib.vtq_debug_attention_variant(-1)
        assert torch.equal(outs[0][:, : nseq * S], outs[1][:, : nseq * S])
        assert torch.equal(outs[0][:, : nseq * S], outs[2][:, : nseq * S])
    # the pipelined kernel's two block walks (XCD-strided, the default; consecutive / paired, vtq_debug_attention_map(1)) and a grid sized
    # for fewer CUs (vtq_debug_cu_partition: a CU-masked stream's share) visit the same blocks: the same bits
    try:
        lib.vtq_debug_attention_variant(1)
        ref = _attention_case(lib, fmt, nseq, S, H, True, spike_row=S - 3).view(torch.int16).clone()
        _lib.check(lib.vtq_debug_attention_map(1))
        walk = _attention_case(lib, fmt, nseq, S, H, True, spike_row=S - 3).view(torch.int16).clone()
        _lib.check(lib.vtq_debug_attention_map(0))
        _lib.check(lib.vtq_debug_cu_partition(0, 64))
        part = _attention_case(lib, fmt, nseq, S, H, True, spike_row=S - 3).view(torch.int16).clone()
    finally:
        lib.vtq_debug_attention_variant(-1)
        lib.vtq_debug_attention_map(0)
        lib.vtq_debug_cu_partition(0, 0)
    assert torch.equal(ref[:, : nseq * S], walk[:, : nseq * S]) and torch.equal(ref[:, : nseq * S], part[:, : nseq * S])


def test_cu_partition_hooks():
    """The measurement hooks behind tools/cu_partition.py: vtq_debug_cu_map reports one distinct CU per CU-filling workgroup on an unmasked stream
    (XCC id 0 .. 7, 32 each on an MI355X), and a persistent GEMM sized for 24 of the 32 CUs of every XCD (its tile schedule rebuilt for 192
    workgroups) gives the bits of the full grid."""
    lib = _lib.load()
    n = torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(2 * n, dtype=torch.int32, device=DEV)
    _lib.check(lib.vtq_debug_cu_map(out.data_ptr(), n, 300, stream()))
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32").reshape(n, 2)
    cus = {(int(x) & 15, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 15) for x, h in v}
    assert len(cus) == n and {c[0] for c in cus} == set(range(8))
    M, N, K, fmt = 4096, 768, 768, "fp16x3"
    A, W, bias = _randn(M, K, seed=5), _randn(N, K, seed=6, scale=0.03), _randn(N, seed=7)
    Ap, Wp = to_planes(A, fmt, "a"), to_planes(W, fmt, "w")
    outs = []
    try:
        _lib.check(lib.vtq_debug_gemm_variant(0))              # the persistent 256x256 kernel
        for g in (0, 24, 20):
            _lib.check(lib.vtq_debug_cu_partition(g, 0))
            o = torch.zeros((Ap.shape[0], M, N), dtype=elt_dtype(fmt), device=DEV)
            _lib.check(lib.vtq_k_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, num_code(fmt), 0, bias.data_ptr(), None, None,
                                      o.data_ptr(), M * N, N, stream()))
            torch.cuda.synchronize()
            outs.append(o.view(torch.int16).clone())
    finally:
        lib.vtq_debug_cu_partition(0, 0)
        lib.vtq_debug_gemm_variant(-1)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert lib.vtq_debug_cu_partition(33, 0) != 0 and lib.vtq_debug_attention_map(2) != 0


@pytest.mark.parametrize("fmt", ["bf16x3", "bf16"])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_huge_logits(fmt, variant):
    """Scores of 1e12 .. 1e13 (Q, K of a 1e7-gain LayerNorm: the fp32 reference's softmax is finite there, train.py:602-607).  The
    exponent must subtract the row maximum exactly: exp2(s c - m c) as one FMA subtracts the ROUNDED product m c and returned inf
    (round 3).  3-term formats take the log2-domain path (scale folded into Q), single-plane formats the magnitude guard."""
    lib = _lib.load()
    nseq, S, H = 3, 200, 768
    rows = nseq * S + 128
    qkv = _randn(rows, 3 * H, seed=21, scale=1.0)
    qkv[:, : 2 * H] *= 2.0e6                                   # Q and K; V stays O(1)
    P = to_planes(qkv, fmt, "a")
    out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device=DEV)
    lib.vtq_debug_attention_variant(variant)
    try:
        _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, nseq, S, S, H, num_code(fmt), stream()))
        torch.cuda.synchronize()
    finally:
        lib.vtq_debug_attention_variant(-1)
    got = planes_value(out)[: nseq * S].view(nseq, S, H)
    assert bool(torch.isfinite(got).all())
    ref = _attention_ref(planes_value(P)[: nseq * S], nseq, S, S, H)      # fp64 softmax of the same (rounded) operands: one-hot rows
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < {"bf16x3": 2e-4, "bf16": 1.5e-2}[fmt], err


@pytest.mark.parametrize("fmt", ["fp16", "bf16"])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_attention_rows_do_not_depend_on_the_next_sequence(fmt, variant):
    """Packed sequences (pitch = S): the lanes of a ragged last query group hold the NEXT sequence's rows.  The single-plane formats'
    magnitude guard (|m c| > 64: subtract the maximum first) is a per-row decision -- a wave-wide one let a huge neighbour switch the
    valid rows of that group to the other rounding path (found by tools/attn_stress.py in round 3: the two kernels, which fill those
    lanes differently, differed in single elements).  Sequence 0 alone and sequence 0 followed by a sequence with huge logits must
    give the same bits for sequence 0."""
    lib = _lib.load()
    S, H = 300, 768                                            # 300 = 9 x 32 + 12: rows 288 .. 299 share a wave with 20 foreign rows
    rows = 2 * S + 128
    qkv = _randn(rows, 3 * H, seed=33, scale=2.0)              # |m c| of a few tens: below the guard for most rows, around it for some
    qkv[S:2 * S, : 2 * H] *= 40.0                              # the neighbour: far beyond it
    outs = []
    for second in (False, True):
        x = qkv.clone()
        if not second:
            x[S:] = 0.0
        P = to_planes(x, fmt, "a")
        out = torch.zeros((P.shape[0], rows, H), dtype=elt_dtype(fmt), device=DEV)
        lib.vtq_debug_attention_variant(variant)
        try:
            _lib.check(lib.vtq_k_attention(P.data_ptr(), rows * 3 * H, out.data_ptr(), rows * H, 2 if second else 1, S, S, H, num_code(fmt), stream()))
            torch.cuda.synchronize()
        finally:
            lib.vtq_debug_attention_variant(-1)
        outs.append(out[:, :S].view(torch.int16).clone())
    assert torch.equal(outs[0], outs[1])


def _skinny(x, W, bias, fmt, epi=0, post=None, gamma=None, res=None, aux=None, nsplit=0, want_y=True, ycols=None, planes_out=False,
            pcol0=0, next_slope=None, Kp=None):
    """Drive vtq_k_skinny_linear: x [R, K] fp32, W [N, K] fp32 -> (y fp32 or None, planes value fp64 or None)."""
    lib = _lib.load()
    R, K = x.shape
    N = W.shape[0]
    Kp = Kp or K
    Ra, Np = (R + 63) // 64 * 64, (N + 15) // 16 * 16
    xpad = torch.zeros(Ra, Kp, device=DEV); xpad[:R, :K] = x
    wpad = torch.zeros(Np, Kp, device=DEV); wpad[:N, :K] = W
    xa, wp = to_planes(xpad, fmt, "a"), to_planes(wpad, fmt, "w")
    y = torch.full((R, N), float("nan"), device=DEV) if want_y else None
    npl = planes_of(fmt, "a")
    ncol = N - pcol0
    ya = torch.zeros((npl, Ra, ncol), dtype=elt_dtype(fmt), device=DEV) if planes_out else None
    ptr = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.vtq_k_skinny_linear(xa.data_ptr(), Ra * Kp, Kp, wp.data_ptr(), Np * Kp, R, N, Kp, num_code(fmt), epi, bias.data_ptr(),
                                       ptr(post), ptr(gamma), ptr(res), ptr(aux), N, nsplit, ptr(y), N, N if ycols is None else ycols,
                                       ptr(ya), Ra * ncol, ncol, pcol0, ptr(next_slope), stream()))
    torch.cuda.synchronize()
    return y, (planes_value(ya)[:R] if planes_out else None), planes_value(xa)[:R, :K], planes_value(wp)[:N, :K]


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("R,N,K", [(2, 768, 768), (32, 864, 768), (64, 3072, 768), (64, 768, 3072), (37, 192, 768), (130, 768, 96), (5, 1, 192)])
def test_skinny_linear_plain(fmt, R, N, K):
    x, W, bias = _randn(R, K, seed=20), _randn(N, K, seed=21, scale=0.05), _randn(N, seed=22)
    y, _, xv, wv = _skinny(x, W, bias, fmt)
    ref = xv @ wv.t() + bias.double()
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("fmt", ["fp16x3", "bf16x3", "fp16x2"])
def test_skinny_linear_epilogues(fmt):
    """Every epilogue form of the CLS tail and the DiffNet head, fp32 and plane outputs, K zero-padded to the k-step (hid = 48)."""
    R, H, hid = 40, 768, 48
    slope, nxt = torch.tensor([0.23], device=DEV), torch.tensor([0.31], device=DEV)
    x, W, bias = _randn(R, H, seed=30), _randn(H, H, seed=31, scale=0.05), _randn(H, seed=32)
    res, aux, gamma = _randn(R, H, seed=33), _randn(R, H, seed=34), _randn(H, seed=35)
    prelu = lambda v, a: torch.where(v >= 0, v, a * v)
    tolp = OUT_TOL[fmt]
    # GELU -> planes only (tail fc1)
    _, pv, xv, wv = _skinny(x, W, bias, fmt, epi=1, want_y=False, planes_out=True)
    pre = xv @ wv.t() + bias.double()
    ref = torch.nn.functional.gelu(pre)
    assert (pv - ref).abs().max().item() < tolp * ref.abs().max().item()
    # PReLU(post) -> planes (q_predictor.1 + .2)
    _, pv, _, _ = _skinny(x, W, bias, fmt, epi=2, post=slope, want_y=False, planes_out=True)
    ref = prelu(pre, 0.23)
    assert (pv - ref).abs().max().item() < tolp * ref.abs().max().item()
    # residual with LayerScale, in place semantics (tail out-proj / fc2) + planes with the consumer's PReLU (RG tail conv)
    y, pv, _, _ = _skinny(x, W, bias, fmt, epi=3, gamma=gamma, res=res, planes_out=True, next_slope=nxt)
    ref = res.double() + gamma.double() * pre
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    assert (pv - prelu(ref, 0.31)).abs().max().item() < tolp * ref.abs().max().item()
    # gate: res + aux * sigmoid(v), K = hid padded to 64 (RCAB stage B)
    t, Wu, bu = _randn(R, hid, seed=36).abs(), _randn(H, hid, seed=37, scale=0.1), _randn(H, seed=38)
    y, pv, tv, wuv = _skinny(t, Wu, bu, fmt, epi=4, res=res, aux=aux, planes_out=True, Kp=64)
    ref = res.double() + aux.double() * torch.sigmoid(tv @ wuv.t() + bu.double())
    assert (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    assert (pv - ref).abs().max().item() < tolp * ref.abs().max().item()
    # convcat: columns < H fp32, columns >= H relu -> planes (RCAB stage A with the CA squeeze folded in)
    Wc, bc = _randn(H + hid, H, seed=39, scale=0.05), _randn(H + hid, seed=40)
    y, pv, xv, wcv = _skinny(x, Wc, bc, fmt, epi=5, nsplit=H, ycols=H, planes_out=True, pcol0=H)
    full = xv @ wcv.t() + bc.double()
    assert (y[:, :H].double() - full[:, :H]).abs().max().item() < 2e-5 * full.abs().max().item()
    assert torch.isnan(y[:, H:]).all()                                      # fp32 columns >= ycols are not written
    assert (pv - torch.relu(full[:, H:])).abs().max().item() < tolp * full.abs().max().item()


@pytest.mark.parametrize("kw,HB", [(dict(), 32), (dict(ca_reduction=16, num_rgs=2, num_rcabs=3), 5), (dict(calibrate=False), 7),
                                   (dict(vit_config=dict(variant="ViT-L16", num_keep_layers=1)), 130)])
@pytest.mark.parametrize("precision", ["fp16x3", "bf16"])
def test_diffnet_head_against_oracle(kw, HB, precision):
    """vtq_k_diffnet_head (quality_decoder -> q_predictor as skinny MFMA stages, fp16 hi/lo whatever the encoder's precision)
    against oracle.quality_decoder / q_predictor on the same CLS differences."""
    import ctypes as C
    import json
    from oracle import vtamiq_oracle as O
    from vtamiq_amd import VTAMIQ, synth
    kw = dict(kw)
    kw.setdefault("vit_config", dict(variant="ViT-B16", num_keep_layers=1))
    kw["vit_config"]["pretrained"] = False
    m = VTAMIQ(**json.loads(json.dumps(kw)), precision=precision)
    sd = synth.make_state_dict(m.spec, 61)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    H = m.spec.hidden_size
    p = torch.zeros(1, 4, 3, 16, 16, device=DEV)
    with torch.no_grad():
        m((p, p), (torch.zeros(1, 4, 2, device=DEV),) * 2, ((torch.zeros(1, 4, device=DEV),) * 2 if m.spec.use_scale_embedding else (None, None)))
    d = _randn(HB, H, seed=62, scale=0.5)
    q = torch.zeros(HB, device=DEV)
    _lib.check(_lib.load().vtq_k_diffnet_head(m._engine, d.data_ptr(), HB, q.data_ptr(), stream()))
    torch.cuda.synchronize()
    t = O.to_torch(sd)
    ref = O.q_predictor(t, O.quality_decoder(t, m.spec, d.cpu())).numpy()
    err = (q.cpu().numpy() - ref)
    assert abs(err).max() < 2e-5 * max(1.0, abs(ref).max()), (abs(err).max(), abs(ref).max())


@pytest.mark.parametrize("P", [16, 8])
@pytest.mark.parametrize("tag", ["aligned", "unaligned"])
def test_device_patch_extraction_bit_exact(tag, P):
    """uint8 images + coordinates -> (patches, pos, scales) on the GPU, bit-exact against the golden captured
    from the reference's get_iqa_patches (3 scales, flips in the unaligned case) and against the oracle."""
    from tests.test_patch_oracle import load_patch_golden
    from vtamiq_amd.patches import extract_patches
    import numpy as np
    g, imgs, flips, samples, dims = load_patch_golden(tag, P)
    nsc = len(samples)
    smp = np.stack([np.concatenate([samples[s][k].T for s in range(nsc)]) for k in range(2)]).astype(np.int32)      # [2, N, 2]
    sid = np.stack([np.concatenate([np.full(samples[s][k].shape[1], s) for s in range(nsc)]) for k in range(2)]).astype(np.int32)
    fl = torch.tensor([[int(flips[0]), int(flips[1])]] * 2, dtype=torch.int32)
    patches, pos, scales = extract_patches(torch.from_numpy(np.stack(imgs)).to(DEV), torch.from_numpy(smp), torch.from_numpy(sid), nsc, fl, patch_size=P)
    torch.cuda.synchronize()
    assert np.array_equal(patches.cpu().numpy(), g[f"{tag}/patches"])
    assert np.array_equal(pos.cpu().numpy(), g[f"{tag}/pos"])
    assert np.array_equal(scales.cpu().numpy(), g[f"{tag}/scales"].astype(np.float32))

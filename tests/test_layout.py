"""CPU-side checks: the C-ABI library builds, loads and exports every declared symbol; the product package never
touches the oracle or the reference; the drop-in module exposes the reference's state_dict layout."""
import ast
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vtamiq_amd import _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "vtamiq_hip.h")).read()
    declared = set(re.findall(r"\b(vtq_[a-z0-9_]+)\s*\(", header))
    declared -= {"vtq_engine"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    lib.vtq_abi_version.restype = ctypes.c_int
    assert lib.vtq_abi_version() == _lib.ABI_VERSION
    # the fp8 experiment has a header of its own and is NOT in the product library ...
    fp8_header = open(os.path.join(ROOT, "include", "vtamiq_hip_fp8.h")).read()
    fp8_declared = set(re.findall(r"\b(vtq_[a-z0-9_]+)\s*\(", fp8_header))
    assert fp8_declared == set(_lib.FP8_SIGNATURES), fp8_declared ^ set(_lib.FP8_SIGNATURES)
    assert not (fp8_declared & declared)
    if os.path.basename(_lib.LIB_PATH) == "libvtamiq_hip.so":
        for name in fp8_declared:
            assert not hasattr(lib, name), f"{name} is exported by the product library"
    # ... and a build of the experiment (python -m vtamiq_amd.build --fp8) exports both sets
    if os.environ.get("VTQ_TEST_FP8_BUILD", "1") == "1":
        fp8_lib = ctypes.CDLL(build.build(verbose=False, fp8=True))
        for name in declared | fp8_declared:
            assert hasattr(fp8_lib, name), name


def test_vtq_create_rejects_unknown_option_bits_before_touching_the_device():
    """unknown vtq_config.options bits and a fused-LayerNorm mlp_dim the kernel cannot run are errors at vtq_create (host-side
    validation: no GPU needed), not a bare hipErrorInvalidValue at the first forward."""
    from vtamiq_amd import _lib
    lib = _lib.load()
    base = dict(hidden_size=768, mlp_dim=3072, num_heads=12, num_layers=1, patch_dim=768, pos_grid=24, num_extra_tokens=0, num_scales=0,
                use_layer_scale=0, calibrate=1, diff_scale=1, num_rgs=1, num_rcabs=1, ca_hidden=96, precision=_lib.PREC_FP16X3, num_adapters=0)
    h = ctypes.c_void_p()
    cfg = _lib.VtqConfig(**base, options=64)
    assert lib.vtq_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and b"options" in lib.vtq_last_error()
    cfg = _lib.VtqConfig(**dict(base, precision=_lib.PREC_FP8), options=0)
    if not _lib.has_fp8():
        assert lib.vtq_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and b"experiment" in lib.vtq_last_error()


def test_fp8_is_not_a_mode_of_the_product_model():
    from vtamiq_amd import VTAMIQ, _lib
    with pytest.raises(NotImplementedError, match="EXPERIMENT"):
        VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, pretrained=False), precision="fp8")
    assert not any(hasattr(VTAMIQ, n) for n in ("fp8_scales", "set_fp8_scales", "calibrate_fp8"))
    if os.path.basename(_lib.LIB_PATH) == "libvtamiq_hip.so":
        assert not _lib.has_fp8()


def test_fp8_experiment_loads_through_its_own_handle_in_a_fresh_interpreter():
    """BASELINE configs[4]: `VTAMIQFp8(...)` constructs with NO environment variable -- the experiment's library is a second handle
    (_lib.load_fp8()), the product library stays what `VTAMIQ` uses, and the two coexist in one process."""
    from vtamiq_amd import build
    build.build(verbose=False, fp8=True)
    code = (
        "from vtamiq_amd import VTAMIQ, _lib\n"
        "from vtamiq_amd.experimental_fp8 import VTAMIQFp8\n"
        "kw = dict(vit_config=dict(variant='ViT-B16', num_keep_layers=1, pretrained=False))\n"
        "m8, m = VTAMIQFp8(**kw), VTAMIQ(**kw, precision='fp16x3')\n"
        "a, b = m8._engine_lib(), m._engine_lib()\n"
        "assert a is not b and _lib.has_fp8(a) and not _lib.has_fp8(b)\n"
        "assert a._name.endswith('libvtamiq_hip_fp8.so') and b._name.endswith('libvtamiq_hip.so')\n"
        "assert _lib.load() is b and _lib.load_fp8() is a and m8.precision == 'fp8'\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("VTQ_")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_vit_checkpoint_loads_into_a_model_without_patch_convolution():
    """transformer.py:643-651 `load_from(weights, use_patch_embedding, use_pos_embedding)` skips the tensors of modules the model was built
    without: a JAX ViT checkpoint must load into a pre-embedded-input model (and one without positional table) without 'unexpected keys'."""
    import numpy as np
    from vtamiq_amd import VTAMIQ, weights
    H, L, G = 768, 1, 24
    rs = np.random.RandomState(3)
    w = {"cls": rs.randn(1, 1, H), "embedding/kernel": rs.randn(16, 16, 3, H), "embedding/bias": rs.randn(H),
         "Transformer/posembed_input/pos_embedding": rs.randn(1, G * G + 1, H),
         "Transformer/encoder_norm/scale": rs.randn(H), "Transformer/encoder_norm/bias": rs.randn(H)}
    r = "Transformer/encoderblock_0"
    for n in ("query", "key", "value"):
        w[f"{r}/MultiHeadDotProductAttention_1/{n}/kernel"] = rs.randn(H, 12, 64)
        w[f"{r}/MultiHeadDotProductAttention_1/{n}/bias"] = rs.randn(12, 64)
    w[f"{r}/MultiHeadDotProductAttention_1/out/kernel"] = rs.randn(12, 64, H)
    w[f"{r}/MultiHeadDotProductAttention_1/out/bias"] = rs.randn(H)
    w[f"{r}/MlpBlock_3/Dense_0/kernel"], w[f"{r}/MlpBlock_3/Dense_0/bias"] = rs.randn(H, 4 * H), rs.randn(4 * H)
    w[f"{r}/MlpBlock_3/Dense_1/kernel"], w[f"{r}/MlpBlock_3/Dense_1/bias"] = rs.randn(4 * H, H), rs.randn(H)
    for n in ("LayerNorm_0", "LayerNorm_2"):
        w[f"{r}/{n}/scale"], w[f"{r}/{n}/bias"] = rs.randn(H), rs.randn(H)
    w = {k: v.astype(np.float32) for k, v in w.items()}
    for extra in (dict(use_patch_embedding=False), dict(use_pos_embedding=False), dict(use_patch_embedding=False, use_pos_embedding=False), {}):
        m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=L, pretrained=False, **extra), precision="fp16x3")
        weights.load_vit_npz(m, w)
        sd = m.state_dict()
        assert ("transformer.embeddings.patch_embeddings.weight" in sd) == extra.get("use_patch_embedding", True)
        assert torch.equal(sd["transformer.encoder.encoder_norm.weight"], torch.from_numpy(w["Transformer/encoder_norm/scale"]))


def test_config_struct_matches_header():
    from vtamiq_amd import _lib
    assert ctypes.sizeof(_lib.VtqConfig) == 20 * 4
    assert _lib.VtqConfig.options.offset == 16 * 4            # include/vtamiq_hip.h: the field behind num_adapters
    assert ctypes.sizeof(_lib.VtqTensorDesc) == 24


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, "vtamiq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            tree = ast.parse(src)
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                for n in names:
                    assert not n.startswith("oracle"), (f, n)
            assert "/root/reference" not in src, f
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read().replace('os.path.isdir("/root/reference")', "")


def test_forward_fails_loudly_without_gpu():
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1), precision="bf16").eval()
    p, pos = torch.zeros(1, 4, 3, 16, 16), torch.zeros(1, 4, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m((p, p), (pos, pos), (None, None))


@pytest.mark.parametrize("kw,ntensors,nparams", [
    (dict(vit_config=dict(variant="ViT-B16", num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True, path_drop_prob=0.1,
                          num_scales=0), ca_reduction=16, rg_path_drop=0.1, predictor_dropout=0.1), 243, 57_322_386),
    (dict(vit_config=dict(variant="ViT-B16")), 326, 101_014_674),
])
def test_state_dict_layout(kw, ntensors, nparams):
    """Key names/shapes of SURVEY.md 8(b); counts probed on the reference (243 tensors / 57.3 M at config defaults)."""
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(**json.loads(json.dumps(kw)))
    sd = m.state_dict()
    layout = m.spec.state_layout()
    assert sorted(sd) == sorted(k for k, _, _ in layout)
    for k, shape, _ in layout:
        assert tuple(sd[k].shape) == tuple(shape), k
    assert len(sd) == ntensors and sum(v.numel() for v in sd.values()) == nparams
    assert len(m.transformer.encoder.layers) == m.spec.num_layers      # train.py:691 reads this


def test_rejected_configs():
    from vtamiq_amd import VTAMIQ
    with pytest.raises(ValueError):
        VTAMIQ(vit_config=dict(variant="ViT-B16", use_cls_token=False))     # crashes in the reference too
    with pytest.warns(UserWarning, match="return_layers"):
        VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, return_attention=True, pretrained=False))   # accepted: VTAMIQ.forward discards them
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, num_adapters=2, pretrained=False))   # adapters: supported (pair 0 applied)
    assert m.spec.num_adapters == 2 and sum("adapter" in k for k in m.state_dict()) == 16
    with pytest.raises(ValueError):
        VTAMIQ(vit_config=dict(variant="ViT-H14"))


def test_model_without_patch_embedding_has_no_convolution():
    """use_patch_embedding=False (backbone.py:20, transformer.py:473-480): no Conv2d, so no patch_embeddings keys; 5-D input then fails like the
    reference's missing attribute (the accelerated path takes pre-embedded (B, N, H) rows: tests/test_gpu_parity.py)."""
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, use_patch_embedding=False, pretrained=False))
    assert not m.spec.use_patch_embedding and not any("patch_embeddings" in k for k in m.state_dict())
    assert sorted(m.state_dict()) == sorted(k for k, _, _ in m.spec.state_layout())
    m.set_freeze_state(True, dict(freeze_dict_vit=None, freeze_quality_decoder=True, freeze_q_predictor=True))


def test_model_without_positional_embedding_has_no_table():
    """use_pos_embedding=False (backbone.py:21, transformer.py:497-499): no UvPosEmbedding module, so no key for it in the state_dict
    -- a checkpoint of such a reference model loads strictly -- and set_freeze_state steps over it like the reference's try/except."""
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, use_pos_embedding=False, pretrained=False))
    assert not m.spec.use_pos_embedding and not any("positional" in k for k in m.state_dict())
    assert sorted(m.state_dict()) == sorted(k for k, _, _ in m.spec.state_layout())
    m.set_freeze_state(True, dict(freeze_dict_vit=None, freeze_quality_decoder=True, freeze_q_predictor=True))
    assert not any(p.requires_grad for p in m.transformer.parameters())


def test_flop_model_matches_baseline_md():
    from vtamiq_amd.spec import make_spec
    s = make_spec(dict(variant="ViT-B16"))
    assert abs(s.flops_per_pair(500) / 1.8992e11 - 1) < 1e-3
    assert abs(s.flops_per_pair(50) / 1.7664e10 - 1) < 1e-3
    d = make_spec(dict(variant="ViT-B16", num_keep_layers=6, num_extra_tokens=8), ca_reduction=16)
    assert abs(d.flops_per_pair(500) / 9.7221e10 - 1) < 1e-3
    l = make_spec(dict(variant="ViT-L16", num_scales=3))
    assert abs(l.flops_per_pair(1024) / 1.4480e12 - 1) < 1e-3


def test_predict_pairwise_branch_matches_reference_semantics():
    """train.py:281-301 on CPU with a stand-in model: two calls sharing `ref`, PreferenceModule = sigmoid(p (q2 - q1)),
    and the fallback sigmoid(q1 - q2) (opposite sign convention, reproduced not fixed)."""
    from vtamiq_amd.predict import predict, PreferenceModule
    B, N = 3, 5
    g = torch.Generator().manual_seed(0)
    patches = torch.randn(B, 3, N, 3, 16, 16, generator=g)
    pos = torch.rand(B, 3, N, 2, generator=g)
    scales = torch.zeros(B, 3, N)
    calls = []

    def model(p, ps, sc):
        calls.append((p, ps, sc))
        return (p[0].mean(dim=(1, 2, 3, 4)) - p[1].mean(dim=(1, 2, 3, 4)), None)
    q, q_p, feats = predict(model, PreferenceModule([2.0]), (torch.zeros(B), patches, pos, scales), True, False, False)
    assert len(calls) == 2 and calls[0][2] == (None, None) and torch.equal(calls[0][0][0], calls[1][0][0])
    q1 = patches[:, 0].mean(dim=(1, 2, 3, 4)) - patches[:, 1].mean(dim=(1, 2, 3, 4))
    q2 = patches[:, 0].mean(dim=(1, 2, 3, 4)) - patches[:, 2].mean(dim=(1, 2, 3, 4))
    assert torch.allclose(q_p, torch.sigmoid(2.0 * (q2 - q1)))
    _, q_p2, _ = predict(model, None, (torch.zeros(B), patches, pos, scales), True, False, False)
    assert torch.allclose(q_p2, torch.sigmoid(q1 - q2))


def test_attention_kernel_rule():
    """Which fused-attention kernel launch_attention picks (attention.hip attention_rule; pure host code): the persistent pipelined
    kernel only for the 3-term formats, only when its 256-row blocks fill >= 74 % of the slots of the persistent grid and pad <= 15 %
    more query rows than 128-row blocks would -- the shapes of profiles/r03_attention_anatomy.txt on a 256-CU device."""
    from vtamiq_amd import _lib
    lib = _lib.load()
    rule = lambda nseq, S, H, fmt, cus=256: lib.vtq_k_attention_rule(nseq, S, H, _lib.NUM[fmt], cus)
    assert rule(64, 501, 768, "fp16x3") == 1 and rule(64, 501, 768, "bf16x3") == 1          # BASELINE configs[1], B = 32
    assert rule(64, 501, 768, "fp16") == 0 and rule(64, 501, 768, "bf16") == 0              # single-plane formats: 4-wave kernel
    assert rule(32, 1025, 1024, "fp16x3") == 1                                              # configs[3]: five 256-row blocks, the fifth with one active wave (the split form lost, round 6)
    assert rule(8, 2501, 768, "fp16x3") == 1                                                # N = 2500: 960 blocks = 3.75 per CU
    assert rule(2, 257, 768, "fp16x3") == 0 and rule(64, 330, 768, "fp16x3") == 0           # 33 % more padded rows
    assert rule(4, 300, 768, "fp16x3") == 0 and rule(3, 51, 768, "fp16x3") == 0             # grids that leave CUs idle
    assert rule(7, 501, 768, "fp16x3") == 0 and rule(8, 501, 768, "fp16x3") == 1            # 168 / 192 blocks of 256 slots: 66 % / 75 %
    assert rule(14, 501, 768, "fp16x3") == 0 and rule(19, 501, 768, "fp16x3") == 1          # 336 of 512 slots: 66 %; 456: 89 %
    assert rule(76, 501, 768, "fp16x3", cus=304) == 1 and rule(26, 501, 768, "fp16x3", cus=304) == 0   # 1824 = 6 x 304; 624 of 912 slots: 68 %
    assert rule(64, 501, 768, "fp16x2") == 0 and rule(64, 501, 768, "fp8") in (0, -1)       # no 2-term attention (the fp16x2 ENGINE mode runs fp16x3 attention)
    # the split form (2: pipelined kernel on the full 256-row blocks + the 4-wave kernel on the rows behind them) is a measurement form only since round 6
    assert rule(32, 521, 768, "fp16x3") == 0 and rule(32, 521, 768, "bf16x3") == 0          # the reference-default topology (512 patches + 9 tokens): 768 padded rows against 640
    assert rule(32, 521, 768, "fp16") == 0 and rule(2, 521, 768, "fp16x3") == 0
    assert rule(32, 577, 768, "fp16x3") == 0 and rule(32, 512, 768, "fp16x3") == 1
    assert rule(64, 257, 768, "fp16x3") == 0 and rule(32, 769, 768, "fp16x3") == 1          # one row past 256: 512 padded rows against 384; past 768: 1024 against 896


def test_gemm_tile_rule_is_host_only():
    """Which tile shape launch_gemm picks (gemm.hip gemm_tile_rule; pure host code): the persistent 256x256 kernel from 64 of its tiles up, below
    that 64x64 tiles while they fit one (<= 256) or two (<= 512) co-resident workgroups per CU, 128x128 beyond; never small tiles for fp8."""
    from vtamiq_amd import _lib
    lib = _lib.load()
    rule = lambda M, N, K, fmt="fp16x3": lib.vtq_k_gemm_tile_rule(M, N, K, _lib.NUM[fmt])
    rows = lambda B, S=501: (2 * B * S + 255) // 256 * 256
    assert [rule(rows(B), 768, 3072) for B in (1, 2, 3, 4, 5, 6, 8, 32)] == [1, 2, 3, 3, 3, 0, 0, 0]          # fc2 / out-proj
    assert [rule(rows(B), 2304, 768) for B in (1, 2, 32)] == [3, 0, 0] and [rule(rows(B), 3072, 768) for B in (1, 2)] == [3, 0]
    assert rule(256, 768, 768, "bf16") == 1 and rule(256, 256, 256, "fp16x2") == 1 and rule(1024, 768, 768, "fp8") == 0
    assert lib.vtq_k_gemm_tile_rule(1024, 768, 768, 99) == -1


def test_gemm_tile_schedule_covers_every_tile_once():
    """The host-built persistent schedule of the GEMM (gemm.hip build_schedule): 256 per-workgroup lists; every 256x256 tile
    appears exactly once, either whole or as its top AND bottom half; half tiles close a list, or -- one of them, on the odd XCDs
    (the epilogue stagger) -- open it; lists are balanced; pure host code, callable without a GPU."""
    import ctypes as C
    import numpy as np
    from vtamiq_amd import _lib
    lib = _lib.load()
    assert lib.vtq_k_gemm_schedule(100, 256, 768, 1, None, 0) == -1
    for M, N, K, wpl in [(256, 256, 768, 1), (1024, 768, 768, 2), (16384, 768, 3072, 2), (16384, 2304, 768, 1), (16384, 3072, 768, 2),
                         (32256, 3072, 768, 1), (32768, 2304, 768, 2), (32768, 768, 768, 1), (33024, 1024, 4096, 2), (65536, 4096, 1024, 1)]:
        n = lib.vtq_k_gemm_schedule(M, N, K, wpl, None, 0)
        out = np.full(n, -1, np.int32)
        assert lib.vtq_k_gemm_schedule(M, N, K, wpl, out.ctypes.data_as(C.c_void_p), n) == n
        nwg = 256
        offs, ent = out[: nwg + 1], out[nwg + 1:]
        assert offs[0] == nwg + 1 and offs[nwg] == n and (np.diff(offs) >= 0).all()
        nt = (M // 256) * (N // 256)
        tile, kind = ent >> 2, ent & 3
        assert tile.min() >= 0 and tile.max() < nt and kind.max() <= 2
        whole = np.bincount(tile[kind == 0], minlength=nt)
        top = np.bincount(tile[kind == 1], minlength=nt)
        bot = np.bincount(tile[kind == 2], minlength=nt)
        assert ((whole == 1) & (top == 0) & (bot == 0) | (whole == 0) & (top == 1) & (bot == 1)).all(), (M, N)
        assert len(ent) == nt + int((kind == 1).sum())
        loads = []
        staggered = 0
        for b in range(nwg):
            k = out[offs[b]:offs[b + 1]] & 3
            if len(k):
                body = k[1:] if (k[0] != 0 and (b % 8) % 2 == 1 and len(k) >= 2 and k[1] == 0) else k     # staggered list: its half tile first
                if body is not k:
                    staggered += 1
                    assert (body == 0).all(), "a staggered list holds one half tile"
                first_half = np.argmax(body != 0) if (body != 0).any() else len(body)
                assert (body[first_half:] != 0).all(), "a whole tile after a half tile"
            loads.append(float((k == 0).sum() + 0.57 * (k != 0).sum()))
        assert max(loads) - min(loads) <= 1.0 + 1e-9, (M, N, min(loads), max(loads))
        if (M, N, K) == (32768, 768, 768):
            assert staggered > 0                  # (whole, half) lists on the odd XCDs run the half first
        # workgroup b runs on XCD b % 8: the XCD owns a contiguous run of the row-major tile order
        for x in range(8):
            mine = np.concatenate([out[offs[b]:offs[b + 1]] >> 2 for b in range(x, nwg, 8)])
            if len(mine):
                u = np.unique(mine)
                assert u.max() - u.min() + 1 == len(u)


def test_abi_rejects_bad_arguments_without_touching_a_gpu():
    """Error behaviour of the C ABI (include/vtamiq_hip.h: non-zero return + vtq_last_error text, no exceptions): argument
    validation happens before any HIP call, so it is checkable here."""
    import ctypes as C
    from vtamiq_amd import _lib
    lib = _lib.load()

    class Cfg(C.Structure):
        _fields_ = [(n, C.c_int32) for n in ("hidden_size", "mlp_dim", "num_heads", "num_layers", "patch_dim", "pos_grid",
                    "num_extra_tokens", "num_scales", "use_layer_scale", "calibrate", "diff_scale", "num_rgs", "num_rcabs",
                    "ca_hidden", "precision", "num_adapters")] + [("reserved", C.c_int32 * 4)]
    good = dict(hidden_size=768, mlp_dim=3072, num_heads=12, num_layers=12, patch_dim=768, pos_grid=24, num_extra_tokens=0,
                num_scales=0, use_layer_scale=0, calibrate=1, diff_scale=1, num_rgs=4, num_rcabs=4, ca_hidden=96, precision=1)
    create = lib.vtq_create
    create.argtypes = [C.c_void_p, C.c_void_p]
    h = C.c_void_p()
    assert create(None, C.byref(h)) != 0 and b"null" in lib.vtq_last_error()
    for field, bad, text in [("hidden_size", 512, b"hidden_size"), ("num_heads", 8, b"head_dim"), ("mlp_dim", 1000, b"mlp_dim"),
                             ("patch_dim", 300, b"patch_dim"), ("num_layers", 0, b"topology"), ("ca_hidden", 3, b"DiffNet"),
                             ("precision", 7, b"precision")]:
        cfg = Cfg(**{**good, field: bad})
        assert create(C.byref(cfg), C.byref(h)) != 0, field
        assert text in lib.vtq_last_error(), (field, lib.vtq_last_error())
    # null handle / null tensors
    assert lib.vtq_forward(None, None, None, None, None, None, None, 1, 1, None, None) != 0
    assert lib.vtq_load_weights(None, None, 0, None) != 0
    assert lib.vtq_reserve(None, 1, 1) != 0
    assert lib.vtq_workspace_bytes(None, 1, 1) == 0
    assert lib.vtq_k_gemm_schedule(256, 100, 768, 1, None, 0) == -1


def test_missing_pretrained_checkpoint_raises(monkeypatch, tmp_path):
    """pretrained=True (the reference's default, backbone.py:25) with no checkpoint file: FileNotFoundError like np.load in the
    reference (transformer.py:622-624), not silently random weights; VTAMIQ_ALLOW_MISSING_WEIGHTS=1 is the explicit opt-out."""
    import warnings
    from vtamiq_amd import VTAMIQ
    monkeypatch.delenv("VTAMIQ_ALLOW_MISSING_WEIGHTS", raising=False)
    monkeypatch.delenv("VTAMIQ_VIT_WEIGHTS", raising=False)
    missing = str(tmp_path / "nope.npz")
    with pytest.raises(FileNotFoundError):
        VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, vit_weights_path=missing))
    VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, vit_weights_path=missing, pretrained=False))     # explicit: fine
    monkeypatch.setenv("VTAMIQ_ALLOW_MISSING_WEIGHTS", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, vit_weights_path=missing))
    assert any("does not exist" in str(x.message) for x in w)


MEASUREMENT_KNOBS = ["VTQ_GEMM_FLAGS", "VTQ_GEMM_CUS", "VTQ_GEMM_CG", "VTQ_GEMM_SCHED", "VTQ_GEMM_STAGGER", "VTQ_ATTN_LDS_PAD",
                     "VTQ_ATTN_VARIANT", "VTQ_NO_CLS_PRUNE", "VTQ_FP8_STATIC_SCALES"]


def test_product_library_reads_no_measurement_environment():
    """the shipped library must not be steerable (or corruptible) through the environment.  Every measurement
    knob is compiled in only with -DVTQ_MEASURE (tools/build_abl.sh); the product objects and the linked library contain neither the
    variable names nor an undefined reference to getenv."""
    import subprocess
    from vtamiq_amd import _lib, build
    build.build(verbose=False)
    blobs = [_lib.LIB_PATH] + [os.path.join(build.OBJ, f) for f in os.listdir(build.OBJ) if f.endswith(".o")]
    assert len(blobs) >= 2
    for path in blobs:
        data = open(path, "rb").read()
        for knob in MEASUREMENT_KNOBS:
            assert knob.encode() not in data, (os.path.basename(path), knob)
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert not re.search(r"\bgetenv\b", nm), "libvtamiq_hip.so imports getenv"
    # and in the sources every getenv of csrc/ goes through the VTQ_MEASURE_ENV gate
    csrc = os.path.join(ROOT, "vtamiq_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            for line in open(os.path.join(csrc, f)):
                if "getenv(" in line:
                    assert "#define VTQ_MEASURE_ENV(name) getenv(name)" in line, (f, line.strip())


def test_weight_signature_sees_replacement_and_new_storage():
    """`p.data = new_tensor`, load_state_dict(assign=True) and module replacement must all change the signature the
    engine's packed weights are keyed by (CPU-only: the signature is host logic)."""
    from vtamiq_amd import VTAMIQ
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, pretrained=False), precision="bf16").eval()
    s0 = m._signature()
    assert m._signature() == s0
    w = m.transformer.encoder.layers[0].attn.query.weight
    w.data = w.data.clone() + 1.0                                   # new storage, version counter untouched
    s1 = m._signature()
    assert s1 != s0
    sd = {k: v.clone() + 0.5 for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                              # new Parameter objects in every submodule
    s2 = m._signature()
    assert s2 != s1
    m.transformer.encoder.layers[0].attn.query = torch.nn.Linear(768, 768)      # nested module replacement
    s3 = m._signature()
    assert s3 != s2
    with torch.no_grad():
        m.q_predictor[1].weight.add_(1.0)                           # in-place write: version counter
    assert m._signature() != s3
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})        # copy_ into the same parameters
    assert m._signature() != s3


def test_weight_change_detection_is_scoped_to_the_model():
    """importing the package registers no process-global torch hook, and modules built elsewhere in the process
    neither invalidate a model's cached parameter walk nor pay for it."""
    import torch.nn.modules.module as M
    before = (len(M._global_parameter_registration_hooks), len(M._global_module_registration_hooks), len(M._global_buffer_registration_hooks))
    from vtamiq_amd import VTAMIQ
    assert (len(M._global_parameter_registration_hooks), len(M._global_module_registration_hooks), len(M._global_buffer_registration_hooks)) == before == (0, 0, 0)
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=1, pretrained=False), precision="bf16").eval()
    s0 = m._signature()
    cache = m.__dict__["_param_cache"]
    other = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.LayerNorm(8))      # an unrelated module: registers parameters and submodules
    other[0].weight = torch.nn.Parameter(torch.zeros(8, 8))
    assert m._signature() == s0 and m.__dict__["_param_cache"] is cache            # same walk object: nothing was invalidated
    m.q_predictor[4] = torch.nn.Linear(192, 1)                                      # a replacement inside THIS model is seen
    assert m._signature() != s0 and m.__dict__["_param_cache"] is not cache


def test_integration_doc_quotes_the_current_abi_version():
    """INTEGRATION.md's stand-alone ctypes stub asserts the ABI version: it must be the header's (it was three versions behind once)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "vtamiq_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    ver = int(re.search(r"#define\s+VTQ_ABI_VERSION\s+(\d+)", hdr).group(1))
    quoted = [int(v) for v in re.findall(r"vtq_abi_version\(\)\s*==\s*(\d+)", doc)]
    assert quoted and all(v == ver for v in quoted), (ver, quoted)

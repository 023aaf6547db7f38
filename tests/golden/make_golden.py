#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE ITSELF (imported from /root/reference, build container only).

The reference's Python never travels to the GPU box; only the small .npz fixtures written here do.
Weights and inputs are regenerated from seeds by vtamiq_amd.synth (our own generator), loaded into the
reference modules through load_state_dict with the reference's key names, and the reference's outputs
are stored.  Absent third-party modules are stubbed in sys.modules (public-API semantics only):
  timm.layers.DropPath / timm.models.layers.{DropPath, trunc_normal_}  (identity in eval)
  cv2, imageio, torchvision, tensorboardX, skimage                       (import-time only, for train.py)

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VTAMIQ_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from vtamiq_amd import synth                      # noqa: E402
from vtamiq_amd.spec import make_spec             # noqa: E402


def _install_stubs():
    class DropPath(torch.nn.Module):               # timm public API: identity unless training with p>0
        def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0. and self.scale_by_keep:
                m.div_(keep)
            return x * m

    timm = types.ModuleType("timm")
    timm.layers = types.ModuleType("timm.layers")
    timm.models = types.ModuleType("timm.models")
    timm.models.layers = types.ModuleType("timm.models.layers")
    timm.layers.DropPath = DropPath
    timm.models.layers.DropPath = DropPath
    timm.models.layers.trunc_normal_ = torch.nn.init.trunc_normal_
    for n, m in (("timm", timm), ("timm.layers", timm.layers), ("timm.models", timm.models),
                 ("timm.models.layers", timm.models.layers)):
        sys.modules[n] = m


def _install_train_stubs():
    """Import-time stubs so that train.py / train_config.py import."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Err(Exception):
        pass
    mod("cv2", error=_Err)
    mod("imageio")
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")
    tv.models = mod("torchvision.models", VGG16_Weights=object, vgg16=None)
    mod("tensorboardX", SummaryWriter=object)
    sk = mod("skimage")
    sk.util = mod("skimage.util")
    sk.util.shape = mod("skimage.util.shape", view_as_windows=None)
    mod("thop", profile=None)


def build_reference(vtamiq_kwargs):
    from modules.vtamiq.vtamiq import VTAMIQ       # reference import
    kw = json.loads(json.dumps(vtamiq_kwargs))     # deep copy (the ctor pops keys)
    kw.setdefault("vit_config", {})["pretrained"] = False
    model = VTAMIQ(**kw)
    model.eval()
    return model


def seeded_state(spec, seed, stress_qk=None, stress_head=False):
    """numpy state dict: the flat random init, or (stress_qk) tests.helpers.stress_state's trained-like statistics (stress_head: with the
    head at a trained model's operating point)."""
    if stress_qk is None:
        return synth.make_state_dict(spec, seed)
    from tests.helpers import stress_state
    return stress_state(spec, seed, qk=float(stress_qk), head=stress_head)


def load_seeded(model, spec, seed, stress_qk=None, stress_head=False):
    sd = {k: torch.from_numpy(v) for k, v in seeded_state(spec, seed, stress_qk, stress_head).items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return sd


def run_case(name, vtamiq_kwargs, B, N, wseed, iseed, aligned=True, trace=False, stress_qk=None, token_num=None, stress_head=False):
    kw = json.loads(json.dumps(vtamiq_kwargs))
    spec = make_spec(**json.loads(json.dumps(kw)))
    if trace:
        kw.setdefault("vit_config", {})["return_layers"] = True
    model = build_reference(kw)
    ref_keys = sorted(model.state_dict().keys())
    our_keys = sorted(k for k, _, _ in spec.state_layout())
    assert ref_keys == our_keys, (set(ref_keys) ^ set(our_keys))
    for k, shape, _ in spec.state_layout():
        assert tuple(model.state_dict()[k].shape) == tuple(shape), (k, shape)
    load_seeded(model, spec, wseed, stress_qk, stress_head)
    patches, pos, scales = synth.make_inputs(spec, B, N, iseed, aligned=aligned)
    tp, tpos = torch.from_numpy(patches), torch.from_numpy(pos)
    p = (tp[:, 0].clone(), tp[:, 1].clone())
    ps = (tpos[:, 0].clone(), tpos[:, 1].clone())
    if scales is not None:
        ts = torch.from_numpy(scales).to(torch.float32)     # train.py:254-255 casts scales to f32
        sc = (ts[:, 0].clone(), ts[:, 1].clone())
    else:
        sc = (None, None)
    out = dict(kwargs=json.dumps(vtamiq_kwargs), B=B, N=N, wseed=wseed, iseed=iseed, aligned=int(aligned))
    if stress_qk is not None:
        out["stress_qk"] = np.float64(stress_qk)
        if stress_head:
            out["stress_head"] = np.int64(1)
    with torch.no_grad():
        q, aux = model(p, ps, sc)
        assert aux is None
        out["q"] = q.numpy().astype(np.float32)
        if token_num is not None:           # the same model with a register token as the IQA token (vtamiq.py:57, 107-108)
            model.token_num = int(token_num)
            out["token_num"] = np.int64(token_num)
            out["q_token"] = model(p, ps, sc)[0].numpy().astype(np.float32)
            model.token_num = 0
        if trace:
            for side, (pp, pq, s_) in (("ref", (p[0], ps[0], sc[0])), ("dist", (p[1], ps[1], sc[1]))):
                x, _, hidden = model.forward_vit(pp, pq, s_, tokens_only=True)
                out[f"tokens_{side}"] = torch.stack(hidden).numpy().astype(np.float32)   # (L,B,T,H), pre final LN
                out[f"final_{side}"] = x.numpy().astype(np.float32)                      # (B,T,H), after encoder_norm
    if stress_qk is not None:
        # the reference evaluated in float64 as well: on these weights two fp32 evaluations of the model differ by ~1e-4, so the
        # fp64 scores are what pins the restatement (oracle in fp64: 1e-13) and what the fp32 reference's own noise is measured against
        m64 = build_reference(kw).double()
        m64.load_state_dict({k: torch.from_numpy(v).double() for k, v in seeded_state(spec, wseed, stress_qk, stress_head).items()}, strict=True)
        with torch.no_grad():
            q64, _ = m64(tuple(t.double() for t in p), tuple(t.double() for t in ps), sc)
        out["q64"] = q64.numpy().astype(np.float64)
    # fingerprints of the regenerated tensors, to detect generator drift
    out["fp_patches"] = np.float64(patches.astype(np.float64).sum())
    out["fp_pos"] = np.float64(pos.astype(np.float64).sum())
    sdnp = seeded_state(spec, wseed, stress_qk, stress_head)
    out["fp_weights"] = np.float64(sum(float(v.astype(np.float64).sum()) for v in sdnp.values()))
    np.savez(os.path.join(HERE, f"{name}.npz"), **out)
    print(f"{name}: q={out['q']}")


def run_ops():
    """Toy-size per-op fixtures with full weights committed."""
    from modules.VisionTransformer import transformer as T
    from modules.RCAN.channel_attention import RCAB, ResidualGroup, CALayer
    g = torch.Generator().manual_seed(7)
    out = {}

    def rnd(*s, scale=1.0):
        return torch.randn(*s, generator=g) * scale

    def randomize(m, scale=0.1):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(rnd(*p.shape, scale=scale) + (1.0 if p.ndim == 1 and p.numel() > 1 and False else 0.0))

    def dump(prefix, m):
        for k, v in m.state_dict().items():
            out[f"{prefix}/sd/{k}"] = v.numpy()

    cfg = {"num_heads": 4, "hidden_size": 64, "mlp_dim": 128, "patch_size": 16, "img_dim": 64}
    with torch.no_grad():
        m = T.MultiHeadSelfAttention(cfg, 1).eval(); randomize(m)
        x = rnd(2, 11, 64)
        y, w = m(x)
        dump("mhsa", m); out["mhsa/x"], out["mhsa/y"], out["mhsa/probs"] = x.numpy(), y.numpy(), w.numpy()

        m = T.MLP(cfg).eval(); randomize(m)
        y = m(x)
        dump("mlp", m); out["mlp/x"], out["mlp/y"] = x.numpy(), y.numpy()

        m = T.EncoderLayer(cfg, True, 0, 1, 0.1).eval(); randomize(m)
        y, _ = m(x)
        dump("layer", m); out["layer/x"], out["layer/y"] = x.numpy(), y.numpy()

        m = T.Embeddings(cfg, True, True, True, 2, 3).eval(); randomize(m)
        pt = rnd(2, 5, 3, 16, 16)
        pos = torch.rand(2, 5, 2, generator=g).clamp(max=1 - 1e-6)
        sc = torch.tensor([[0., 1., 2., 5., -1.], [2., 2., 0., 1., 1.]])
        y = m(pt, pos, sc)
        dump("emb", m)
        out["emb/patches"], out["emb/pos"], out["emb/scales"], out["emb/y"] = pt.numpy(), pos.numpy(), sc.numpy(), y.numpy()

        m = T.UvPosEmbedding({"hidden_size": 8, "patch_size": 16, "img_dim": 384}).eval(); randomize(m)
        pos = torch.tensor([[0., 0.], [0.999999, 0.999999], [0.5, 0.25], [1 / 24., 23 / 24.], [0.0416, 0.9584]])
        out["uvpos/pos"], out["uvpos/y"] = pos.numpy(), m(pos).numpy()
        dump("uvpos", m)

        m = RCAB(64, 8, use_bn=False, input1d=True).eval(); randomize(m)
        x1 = rnd(3, 64, 1)
        dump("rcab", m); out["rcab/x"], out["rcab/y"] = x1.numpy(), m(x1).numpy()

        m = ResidualGroup(64, 2, reduction=8, path_drop_prob=0.1, use_bn=False, input1d=True).eval(); randomize(m)
        dump("rg", m); out["rg/x"], out["rg/y"] = x1.numpy(), m(x1).numpy()
    np.savez(os.path.join(HERE, "ops_toy.npz"), **out)
    print("ops_toy: ", len(out), "arrays")


def run_plumbing():
    """BASELINE config 1 'forward via train.py': get_data_tuple -> predict (non-pairwise).  train.py:254-314."""
    _install_train_stubs()
    import train as ref_train
    kw = dict(vit_config=dict(variant="ViT-B16"))
    spec = make_spec(**json.loads(json.dumps(kw)))
    model = build_reference(kw)
    load_seeded(model, spec, 3)
    B, N = 2, 50
    patches, pos, _ = synth.make_inputs(spec, B, N, 77)
    q_true = np.array([0.25, 0.75], dtype=np.float64)
    scales = np.full((B,), -1, dtype=np.int32)                     # single-scale loader output (patch_datasets.py)
    batch = (torch.from_numpy(q_true), torch.from_numpy(patches), torch.from_numpy(pos), torch.from_numpy(scales))
    with torch.no_grad():
        data = ref_train.get_data_tuple(batch, torch.device("cpu"))
        q, q_p, feats = ref_train.predict(model, None, data, False, False, False)
    assert feats is None
    np.savez(os.path.join(HERE, "plumbing_c1.npz"), kwargs=json.dumps(kw), B=B, N=N, wseed=3, iseed=77,
             q_in=q_true, q=q.numpy(), q_p=q_p.numpy().astype(np.float32))
    print("plumbing_c1:", q.numpy(), q_p.numpy())


def run_npz():
    """JAX .npz ingestion.  A tiny synthetic checkpoint (grid 3x3 -> resized to the model's 4x4) is loaded by the
    reference's VisionTransformer.load_from; inputs and the resulting state_dict are the fixture."""
    from modules.VisionTransformer import transformer as T
    rs = np.random.RandomState(5)
    H, M, L, P = 32, 64, 2, 16
    cfg = dict(vit_weights_path="", img_dim=64, patch_size=P, hidden_size=H, mlp_dim=M, num_heads=4, num_layers=3)
    w = {"embedding/kernel": rs.randn(P, P, 3, H), "embedding/bias": rs.randn(H), "cls": rs.randn(1, 1, H),
         "Transformer/posembed_input/pos_embedding": rs.randn(1, 3 * 3 + 1, H),
         "Transformer/encoder_norm/scale": rs.randn(H), "Transformer/encoder_norm/bias": rs.randn(H)}
    for i in range(3):
        r = f"Transformer/encoderblock_{i}"
        for nm in ("query", "key", "value"):
            w[f"{r}/MultiHeadDotProductAttention_1/{nm}/kernel"] = rs.randn(H, 4, H // 4)
            w[f"{r}/MultiHeadDotProductAttention_1/{nm}/bias"] = rs.randn(4, H // 4)
        w[f"{r}/MultiHeadDotProductAttention_1/out/kernel"] = rs.randn(4, H // 4, H)
        w[f"{r}/MultiHeadDotProductAttention_1/out/bias"] = rs.randn(H)
        w[f"{r}/MlpBlock_3/Dense_0/kernel"] = rs.randn(H, M); w[f"{r}/MlpBlock_3/Dense_0/bias"] = rs.randn(M)
        w[f"{r}/MlpBlock_3/Dense_1/kernel"] = rs.randn(M, H); w[f"{r}/MlpBlock_3/Dense_1/bias"] = rs.randn(H)
        for ln in ("LayerNorm_0", "LayerNorm_2"):
            w[f"{r}/{ln}/scale"] = rs.randn(H); w[f"{r}/{ln}/bias"] = rs.randn(H)
    w = {k: v.astype(np.float32) for k, v in w.items()}
    vit = T.VisionTransformer(cfg, use_classifier=False, num_keep_layers=L, num_extra_tokens=2, pretrained=False)
    vit.load_from(w, True, True)
    out = {"in/" + k: v for k, v in w.items()}
    for k, v in vit.state_dict().items():
        out["sd/transformer." + k] = v.numpy()
    out["meta"] = np.array([H, L, 4 * 4 + 1])
    np.savez(os.path.join(HERE, "npz_ingest_tiny.npz"), **out)
    print("npz_ingest_tiny:", len(out), "arrays")


def run_patches(P=16, fname="patches_gather.npz", seed=11):
    """the gather / position / pyramid part of the loader, pinned by the reference's get_iqa_patches driven by a
    RECORDING sampler (deterministic coordinates; the sampler's RNG is out of scope).  P = 16 (ViT-B16 / L16) and P = 8 (ViT-B8)."""
    _install_train_stubs()
    from data import patch_sampling as PS
    from oracle.patch_oracle import transform_img
    rs = np.random.RandomState(seed)
    H, W = 160, 208
    imgs = [rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8) for _ in range(2)]
    imgs[1] = np.clip(imgs[0].astype(np.int32) + rs.randint(-20, 21, size=(H, W, 3)), 0, 255).astype(np.uint8)

    class Recorder:
        def __init__(self):
            self.calls = []
        def compute_diff(self, imgs):
            return None
        def get_sample_params(self, h, w, ho, wo, diff=None, num_samples=1, debug=False):
            smp = np.stack([rs.randint(0, h - ho + 1, size=num_samples), rs.randint(0, w - wo + 1, size=num_samples)]).astype(np.int64)
            self.calls.append((h, w, smp))
            return smp

    out = {"img0": imgs[0], "img1": imgs[1]}
    for tag, aligned, flips in (("aligned", True, (False, False)), ("unaligned", False, (True, True))):
        rec = Recorder()
        tens = [transform_img(im, flips[0], flips[1]) for im in imgs]
        patches, pos, scales = PS.get_iqa_patches(imgs, tens, 45, P, rec, 3, scale_num_samples_ratio=1.75,
                                                  use_aligned_patches=aligned)
        out[f"{tag}/patches"], out[f"{tag}/pos"], out[f"{tag}/scales"] = patches.numpy(), pos.numpy(), scales.numpy()
        out[f"{tag}/flips"] = np.array(flips, dtype=np.int32)
        out[f"{tag}/ncalls"] = np.array(len(rec.calls))
        for i, (h, w, smp) in enumerate(rec.calls):
            out[f"{tag}/call{i}/hw"] = np.array([h, w]); out[f"{tag}/call{i}/samples"] = smp
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("patches_gather:", {k: v.shape for k, v in out.items() if k.endswith("patches")})


def run_validation_metrics():
    """train.py:398-409 + utils/misc/correlations.py:21-52 on seeded scores: 3 cases (plain, heavy ties, repeats = 1)."""
    _install_train_stubs()
    import train as ref_train
    rng = np.random.default_rng(2024)
    out = {}
    cases = [("mos", 37, 4, 8, 0.1), ("ties", 203, 3, 32, 0.02), ("single", 64, 1, 16, 0.0)]
    for name, n, reps, bs, quant in cases:
        q = rng.uniform(0.0, 1.0, n)
        if quant:
            q = np.round(q / quant) * quant                                   # MOS-like targets with ties
        q = q.astype(np.float32)
        ys, yp, flat_p = [], [], []
        for r in range(reps):
            pred = (0.7 * q + 0.1 + 0.08 * rng.standard_normal(n)).astype(np.float32)
            if name == "ties":
                pred = (np.round(pred * 20) / 20).astype(np.float32)          # ties in the predictions too
            flat_p.append(pred)
            for i in range(0, n, bs):
                ys.append(torch.from_numpy(q[i:i + bs].copy()))
                yp.append(torch.from_numpy(pred[i:i + bs].copy()))
        corr = ref_train.compute_correlations_cat_flat(ys, yp, reps)
        flat_p = np.concatenate(flat_p)
        out[name + "_q"] = q
        out[name + "_pred"] = flat_p                                           # [reps * n], pass-major
        out[name + "_reps"] = reps
        out[name + "_bs"] = bs
        out[name + "_mean"] = ref_train.average_over_repeats(np.array(flat_p, dtype=float), reps) if reps > 1 else np.array(flat_p, dtype=float)
        for k, v in corr.items():
            out[name + "_" + k] = np.float64(v)
        print("validation_metrics", name, {k: float(v) for k, v in corr.items()})
    np.savez(os.path.join(HERE, "validation_metrics.npz"), **out)


def run_ladder_case(name, vtamiq_kwargs, images, N, wseed, iseed, stress_qk, chunk=8, stress_head=False):
    """SCORES ONLY, at the BASELINE patch count: the reference in fp32 and in float64 on stress_state weights over a distortion
    ladder of images * 8 pairs (synth.make_ladder_inputs) -- the trained-like parity tail pinned by the reference itself at
    N = 500.  Run in chunks of `chunk` pairs (the reference materialises every attention matrix)."""
    kw = json.loads(json.dumps(vtamiq_kwargs))
    spec = make_spec(**json.loads(json.dumps(kw)))
    patches, pos, scales = synth.make_ladder_inputs(spec, images, N, iseed)
    assert scales is None
    B = patches.shape[0]
    sdnp = seeded_state(spec, wseed, stress_qk, stress_head)
    out = dict(kwargs=json.dumps(vtamiq_kwargs), images=images, B=B, N=N, wseed=wseed, iseed=iseed, stress_qk=np.float64(stress_qk),
               stress_head=int(stress_head))
    for tag, dt in (("q", torch.float32), ("q64", torch.float64)):
        model = build_reference(kw).to(dt)
        model.load_state_dict({k: torch.from_numpy(v).to(dt) for k, v in sdnp.items()}, strict=True)
        qs = []
        with torch.no_grad():
            for i in range(0, B, chunk):
                tp, tq = torch.from_numpy(patches[i:i + chunk]).to(dt), torch.from_numpy(pos[i:i + chunk]).to(dt)
                q, aux = model((tp[:, 0].clone(), tp[:, 1].clone()), (tq[:, 0].clone(), tq[:, 1].clone()), (None, None))
                assert aux is None
                qs.append(q)
                print(f"  {name} {tag}: {i + chunk}/{B}", flush=True)
        out[tag] = torch.cat(qs).numpy().astype(np.float32 if dt == torch.float32 else np.float64)
        del model
    out["fp_patches"] = np.float64(patches.astype(np.float64).sum())
    out["fp_pos"] = np.float64(pos.astype(np.float64).sum())
    out["fp_weights"] = np.float64(sum(float(v.astype(np.float64).sum()) for v in sdnp.values()))
    np.savez(os.path.join(HERE, f"{name}.npz"), **out)
    d = np.abs(out["q"].astype(np.float64) - out["q64"]) / np.abs(out["q64"])
    print(f"{name}: {B} scores, rms {np.sqrt(np.mean(out['q64'] ** 2)):.4f}; reference fp32 vs its own fp64: max {d.max():.2e}, p95 {np.percentile(d, 95):.2e}")


def run_fullsize():
    """Scores only, at the sizes bench.py runs: BASELINE configs[1] (B = 32, N = 500, ViT-B/16 L = 12) and the reference-default topology row
    (train_config.py:169-194: L = 6, 8 registers, LayerScale, r = 16; B = 16, N = 512), flat seeded weights, through the reference (round 4)."""
    run_case("c2_b32_n500", dict(vit_config=dict(variant="ViT-B16")), B=32, N=500, wseed=51, iseed=61)
    run_case("refdefault_b16_n512",
             dict(vit_config=dict(variant="ViT-B16", num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True, path_drop_prob=0.1, num_scales=0),
                  ca_reduction=16, rg_path_drop=0.1, predictor_dropout=0.1), B=16, N=512, wseed=52, iseed=62)


def run_fullsize_c4():
    """BASELINE configs[3] whole: ViT-L/16 (H = 1024, L = 24, 16 heads), B = 16 pairs, N = 1024 patches over 3 scales (738 / 220 / 66), through
    the reference; scores only (round 4; ~10 min of CPU)."""
    run_case("c4_vitl_b16_n1024", dict(vit_config=dict(variant="ViT-L16", num_scales=3)), B=16, N=1024, wseed=53, iseed=63)


def run_ladder():
    run_ladder_case("stress5_b64_n500", dict(vit_config=dict(variant="ViT-B16")), images=8, N=500, wseed=32, iseed=777, stress_qk=5.0)


def run_operating_point():
    """The 64-pair N = 500 ladder through a head at a trained model's operating point (tests.helpers.stress_state(head=True): scores in
    [0.2, 0.8] instead of cancellation remainders around zero), scored by the reference in fp32 and float64 (round 5; ~10 min of CPU)."""
    run_ladder_case("stress5h_b64_n500", dict(vit_config=dict(variant="ViT-B16")), images=8, N=500, wseed=32, iseed=777, stress_qk=5.0,
                    stress_head=True)


def run_long():
    """The long-sequence regime the reference advertises (README.md:85 "50, 500, and 5000 patches"; data/patch_sampling.py:450): ViT-B/16 L = 12 at
    N = 5000 patches (S = 5001: attention is > 50 % of the flops) through the reference -- flat seeded weights (B = 2, fp32) and trained-like
    statistics with the head at its operating point (B = 1, fp32 + float64); scores only (round 6; ~5 min of CPU, ~8 GB)."""
    run_case("long_b2_n5000", dict(vit_config=dict(variant="ViT-B16")), B=2, N=5000, wseed=54, iseed=64)
    run_case("long5h_b1_n5000", dict(vit_config=dict(variant="ViT-B16")), B=1, N=5000, wseed=55, iseed=65, stress_qk=5.0, stress_head=True)


def run_stress():
    """The reference itself on weights with trained-ViT-like statistics (tests.helpers.stress_state: peaked softmax, outlier
    channels) -- the flat random init of the other cases exercises none of that (transformer.py:153-172)."""
    B16 = "ViT-B16"
    run_case("stress3_b3_n90", dict(vit_config=dict(variant=B16)), B=3, N=90, wseed=31, iseed=41, stress_qk=3.0)
    run_case("stress5_b3_n90", dict(vit_config=dict(variant=B16)), B=3, N=90, wseed=32, iseed=42, stress_qk=5.0)


def main():
    sys.path.insert(0, REF)
    _install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    B16 = "ViT-B16"
    if len(sys.argv) > 1 and sys.argv[1] == "--adapters":       # only the adapter case (added in round 2; the others are unchanged)
        run_case("adapters_b2_n40", dict(vit_config=dict(variant=B16, num_keep_layers=2, num_adapters=2, use_layer_scale=True, num_extra_tokens=1)),
                 B=2, N=40, wseed=22, iseed=19)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--vitb8":          # only the ViT-B/8 case (added in round 2; the others are unchanged)
        run_case("vitb8_b2_n90", dict(vit_config=dict(variant="ViT-B8", num_keep_layers=3, num_scales=2)), B=2, N=90, wseed=8, iseed=18)
        run_patches(P=8, fname="patches_gather_p8.npz", seed=12)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--nopos":          # only the use_pos_embedding=False case (added in round 5)
        run_case("nopos_b2_n40", dict(vit_config=dict(variant=B16, num_keep_layers=2, use_pos_embedding=False, num_extra_tokens=1, num_scales=2)),
                 B=2, N=40, wseed=23, iseed=20)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--preembedded":    # only the use_patch_embedding=False case (added in round 5): (B, N, H) rows as input
        run_case("preemb_b3_n60", dict(vit_config=dict(variant=B16, num_keep_layers=2, use_patch_embedding=False, num_extra_tokens=2, num_scales=3)),
                 B=3, N=60, wseed=29, iseed=23, aligned=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--token":          # only the token_num case (added in round 5): CLS scores + register-token scores
        run_case("token2_b3_n45", dict(vit_config=dict(variant=B16, num_keep_layers=3, num_extra_tokens=3, use_layer_scale=True)),
                 B=3, N=45, wseed=32, iseed=21, token_num=2)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--fullsize":       # only the two full-size score goldens (added in round 4; ~3 min)
        run_fullsize()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--fullsize-c4":
        run_fullsize_c4()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--ladder":         # only the 64-pair N = 500 trained-like case (added in round 4; ~10 min)
        run_ladder()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--operating-point":   # only the operating-point ladder (added in round 5; ~10 min)
        run_operating_point()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--long":           # only the N = 5000 cases (added in round 6)
        run_long()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--stress":         # only the trained-like-statistics cases (added in round 3)
        run_stress()
        return
    run_case("c1_b2_n50", dict(vit_config=dict(variant=B16)), B=2, N=50, wseed=1, iseed=11, trace=True)
    run_case("refdefault_b2_n64",
             dict(vit_config=dict(variant=B16, num_keep_layers=6, num_extra_tokens=8, use_layer_scale=True,
                                  path_drop_prob=0.1, num_scales=0),
                  ca_reduction=16, rg_path_drop=0.1, predictor_dropout=0.1), B=2, N=64, wseed=2, iseed=12)
    run_case("scales3_b2_n40", dict(vit_config=dict(variant=B16, num_keep_layers=2, num_scales=3, num_extra_tokens=2)),
             B=2, N=40, wseed=3, iseed=13)
    run_case("unaligned_b3_n50", dict(vit_config=dict(variant=B16, num_keep_layers=3)), B=3, N=50, wseed=4, iseed=14,
             aligned=False)
    run_case("c2shape_b4_n500", dict(vit_config=dict(variant=B16)), B=4, N=500, wseed=5, iseed=15)
    run_case("vitl_b2_n70", dict(vit_config=dict(variant="ViT-L16", num_keep_layers=2, num_scales=3)),
             B=2, N=70, wseed=6, iseed=16)
    run_case("nocalib_b2_n30", dict(vit_config=dict(variant=B16, num_keep_layers=1), calibrate=False, diff_scale=False),
             B=2, N=30, wseed=7, iseed=17)
    run_case("vitb8_b2_n90", dict(vit_config=dict(variant="ViT-B8", num_keep_layers=3, num_scales=2)), B=2, N=90, wseed=8, iseed=18)
    run_case("adapters_b2_n40", dict(vit_config=dict(variant=B16, num_keep_layers=2, num_adapters=2, use_layer_scale=True, num_extra_tokens=1)),
             B=2, N=40, wseed=22, iseed=19)
    run_case("nopos_b2_n40", dict(vit_config=dict(variant=B16, num_keep_layers=2, use_pos_embedding=False, num_extra_tokens=1, num_scales=2)),
             B=2, N=40, wseed=23, iseed=20)
    run_case("token2_b3_n45", dict(vit_config=dict(variant=B16, num_keep_layers=3, num_extra_tokens=3, use_layer_scale=True)),
             B=3, N=45, wseed=32, iseed=21, token_num=2)
    run_case("preemb_b3_n60", dict(vit_config=dict(variant=B16, num_keep_layers=2, use_patch_embedding=False, num_extra_tokens=2, num_scales=3)),
             B=3, N=60, wseed=29, iseed=23, aligned=False)
    run_stress()
    run_ladder()
    run_operating_point()
    run_fullsize()
    run_fullsize_c4()
    run_ops()
    run_npz()
    run_plumbing()
    run_patches()
    run_patches(P=8, fname="patches_gather_p8.npz", seed=12)
    run_validation_metrics()


if __name__ == "__main__":
    main()

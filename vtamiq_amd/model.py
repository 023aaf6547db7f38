"""Drop-in model callable for the reference's `VTAMIQ` (modules/vtamiq/vtamiq.py:26-119) on MI355X.

Same constructor kwargs, same `state_dict()` key layout, same call
    q, aux = model((p_ref, p_dist), (pos_ref, pos_dist), (sc_ref, sc_dist))      # aux is None, q: (B,) float32
but `forward` runs the hand-written gfx950 engine (libvtamiq_hip.so) through its C ABI.

The torch modules below are PARAMETER CONTAINERS only (they give the reference's key names and shapes and are what
`load_state_dict`, `.to()`, `.parameters()` and `set_freeze_state` operate on); none of them is ever called.
There is no CPU or eager fallback: a forward on CPU tensors, in train mode, or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Optional

import torch
from torch import nn

from . import _lib
from .spec import ModelSpec, make_spec

_PRECISIONS = _lib.PRECISIONS
# "auto" (the default of the drop-in model): run the parity mode fp16x3, read the engine's error word after every forward (one
# 4-byte D2H + stream sync: the caller's q.cpu() / loss pays the same sync a moment later) and
#   * raise IndexError for a position outside [0, 1), as the reference's table lookup does (transformer.py:417-421);
#   * when an operand left the fp16 range (|v| > 65504: non-finite CLS difference) switch THIS model to "bf16x3" (fp32 range,
#     same 3-MFMA split) with a warning and re-run the call -- the fp32 reference returns finite scores there (train.py:602-607),
#     so must a drop-in.  The switch is sticky: the overflow comes from the checkpoint's activation scale.
# An explicit precision keeps the forward asynchronous (bench.py, throughput runs); check_inputs() reports on demand.
DEFAULT_PRECISION = "auto"
AUTO_FIRST, AUTO_FALLBACK = "fp16x3", "bf16x3"

class _Params(nn.Module):
    """A container whose forward must never run."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container of the HIP engine; call the VTAMIQ model instead")


class _Gamma(_Params):                      # LayerScale (transformer.py:235-243)
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))


class _UvPos(_Params):                      # UvPosEmbedding (transformer.py:403-415)
    def __init__(self, n, H):
        super().__init__()
        self.positional_embeddings = nn.Parameter(torch.zeros(1, n, H).normal_(std=0.02))


class _ScaleEmb(_Params):                   # ScaleEmbedding (transformer.py:385-394)
    def __init__(self, n, H):
        super().__init__()
        self.scale_embeddings = nn.Parameter(torch.zeros(1, n, H).normal_(std=0.02))


class _Embeddings(_Params):                 # Embeddings (transformer.py:458-505)
    def __init__(self, spec: ModelSpec):
        super().__init__()
        H, P = spec.hidden_size, spec.patch_size
        if spec.use_patch_embedding:
            self.patch_embeddings = nn.Conv2d(3, H, kernel_size=P, stride=P)
        self.use_patch_embedding = spec.use_patch_embedding
        self.cls_token = nn.Parameter(torch.zeros(1, 1, H).normal_(std=0.02))
        if spec.num_extra_tokens > 0:
            self.extra_tokens = nn.Parameter(torch.zeros(1, spec.num_extra_tokens, H).normal_(std=0.02))
        if spec.use_pos_embedding:
            self.positional_embeddings = _UvPos(spec.pos_grid ** 2 + 1, H)
        if spec.use_scale_embedding:
            self.scale_embeddings = _ScaleEmb(spec.num_scales + 1, H)
        self.num_tokens = spec.num_tokens
        self.use_pos_embedding = spec.use_pos_embedding
        self.use_scale_embedding = spec.use_scale_embedding


class _Attn(_Params):                       # MultiHeadSelfAttention (transformer.py:125-146)
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value, self.out = (nn.Linear(H, H) for _ in range(4))


class _Mlp(_Params):                        # MLP (transformer.py:197-210)
    def __init__(self, H, M):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(H, M), nn.Linear(M, H)


class _Adapter(_Params):                    # Adapter (transformer.py:177-194): x + Linear(H/4 -> H)(GELU(Linear(H -> H/4)(x)))
    def __init__(self, H):
        super().__init__()
        self.adapter = nn.Sequential(nn.Linear(H, H // 4), nn.Identity(), nn.Linear(H // 4, H))


class _EncoderLayer(_Params):               # EncoderLayer (transformer.py:246-273)
    def __init__(self, spec: ModelSpec):
        super().__init__()
        H = spec.hidden_size
        self.attention_norm = nn.LayerNorm(H, eps=1e-6)
        self.ffn_norm = nn.LayerNorm(H, eps=1e-6)
        self.ffn = _Mlp(H, spec.mlp_dim)
        self.attn = _Attn(H)
        self.use_adapters = spec.num_adapters > 0
        for a in range(1, 2 * spec.num_adapters + 1):          # adapter1 / adapter2 = pair 0 (the one the forward uses), ...
            self.add_module(f"adapter{a}", _Adapter(H))
        if spec.use_layer_scale:
            self.ls1, self.ls2 = _Gamma(H), _Gamma(H)


class _Encoder(_Params):                    # Encoder (transformer.py:328-361)
    def __init__(self, spec: ModelSpec):
        super().__init__()
        self.encoder_norm = nn.LayerNorm(spec.hidden_size, eps=1e-6)
        self.layers = nn.ModuleList(_EncoderLayer(spec) for _ in range(spec.num_layers))


class _Transformer(_Params):                # VisionTransformer (transformer.py:565-626)
    def __init__(self, spec: ModelSpec):
        super().__init__()
        self.hidden_size = spec.hidden_size
        self.use_layer_scale = spec.use_layer_scale
        self.use_adapters = spec.num_adapters > 0
        self.embeddings = _Embeddings(spec)
        self.encoder = _Encoder(spec)
        for m in self.modules():            # _init_weights (transformer.py:670-678)
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)


def _ca_layer(H, hid):                      # CALayer.conv_du (channel_attention.py:53-62, 77-80); indices 1 and 4 hold params
    m = _Params()
    m.conv_du = nn.Sequential(nn.Identity(), nn.Conv1d(H, hid, 1), nn.Identity(), nn.Identity(), nn.Conv1d(hid, H, 1),
                              nn.Identity())
    return m


def _rcab(H, hid):                          # RCAB.body (channel_attention.py:41-47); indices 1, 2, 4 hold params
    m = _Params()
    m.body = nn.Sequential(nn.Identity(), nn.PReLU(), nn.Conv1d(H, H, 1), nn.Identity(), _ca_layer(H, hid))
    return m


def _residual_group(H, hid, num_rcabs):     # ResidualGroup.body (channel_attention.py:18-25)
    m = _Params()
    m.body = nn.Sequential(*[_rcab(H, hid) for _ in range(num_rcabs)], nn.Conv1d(H, H, 1))
    return m


class VTAMIQ(nn.Module):
    _FP8_EXPERIMENT = False                       # vtamiq_amd.experimental_fp8.VTAMIQFp8 sets it

    def __init__(self, vit_config=None, calibrate=True, diff_scale=True, num_rgs=4, num_rcabs=4, rg_path_drop=0.1,
                 ca_reduction=8, predictor_dropout=0., return_features=False, precision: Optional[str] = None,
                 engine_options: int = 0, **kwargs):
        super().__init__()
        self.engine_options = int(engine_options)      # vtq_config.options (_lib.OPT_*): tests and A/B measurement; 0 = product path
        for k, v in kwargs.items():         # reference only warns about unknown kwargs (vtamiq.py:49)
            warnings.warn(f"[VTAMIQ] Unused kwarg [{k}={v}]")
        vit_config = dict(vit_config or {})
        # backbone.py:25: `pretrained` defaults to True; transformer.py:621-624 then np.load()s config["vit_weights_path"]
        pretrained = bool(vit_config.get("pretrained", True))
        weights_path = vit_config.pop("vit_weights_path", None)
        self.spec = make_spec(vit_config, calibrate=calibrate, diff_scale=diff_scale, num_rgs=num_rgs,
                              num_rcabs=num_rcabs, ca_reduction=ca_reduction)
        spec = self.spec
        H = spec.hidden_size
        self.transformer = _Transformer(spec)
        self.token_num = 0                                                   # vtamiq.py:57
        self.diff_scale = _Gamma(H) if diff_scale else nn.Sequential()       # vtamiq.py:61
        if calibrate:                                                        # vtamiq.py:63-69
            self.quality_decoder = nn.Sequential(
                *[_residual_group(H, spec.ca_hidden, num_rcabs) for _ in range(num_rgs)], nn.Conv1d(H, H, 1))
        else:
            self.quality_decoder = nn.Sequential()
        self.rg_path_drop = rg_path_drop          # stochastic only in train mode, which this path rejects
        self.predictor_dropout = predictor_dropout
        self.q_predictor = nn.Sequential(nn.Identity(), nn.Linear(H, H // 4), nn.PReLU(), nn.Identity(),
                                         nn.Linear(H // 4, 1))              # vtamiq.py:71-77 (Dropout slots 0 and 3)
        self.return_features = return_features
        precision = precision or os.environ.get("VTAMIQ_PRECISION", DEFAULT_PRECISION)
        if precision == "fp8w":                 # alias used by the round-1 review for BASELINE configs[4] ("fp8 weights"): same mode
            precision = "fp8"
        if precision != "auto" and precision not in _PRECISIONS:
            raise ValueError(f"precision must be 'auto' or one of {sorted(_PRECISIONS)}, got {precision!r}")
        if precision == "fp8" and not self._FP8_EXPERIMENT:
            raise NotImplementedError("precision 'fp8' is a throughput EXPERIMENT, not a scoring mode (SROCC 0.66 - 0.84 against the fp32 scores, "
                                      "profiles/r04_fp8_study.txt), and is not part of this model class or of the shipped library: "
                                      "vtamiq_amd.experimental_fp8.VTAMIQFp8 on a library built with `python -m vtamiq_amd.build --fp8`")
        self.precision = precision
        self._auto_fallback = False             # "auto" only: an fp16 operand overflowed, the engine now runs AUTO_FALLBACK
        self._engine = None
        self._engine_device = None
        self._engine_precision = None
        self._weights_sig = None
        self._warned_grad = False
        # VTAMIQ_VALIDATE_INPUTS=1 (or model.validate_inputs = True): synchronise after every forward and raise IndexError when a
        # position lay outside [0, 1), as the reference's table lookup does (transformer.py:417-421).  Off by default: the
        # forward stays asynchronous and such an index is clamped into the table (never read out of bounds); check_inputs()
        # reports it on demand.
        self.validate_inputs = os.environ.get("VTAMIQ_VALIDATE_INPUTS", "0") == "1"
        if pretrained:
            from .weights import default_vit_weights_path, load_vit_npz
            path = weights_path or os.environ.get("VTAMIQ_VIT_WEIGHTS") or default_vit_weights_path(self.spec.variant)
            if os.path.exists(path):
                print("ViT: Loading pretrained transformer from path:", path)          # transformer.py:623
                load_vit_npz(self, path)
            elif os.environ.get("VTAMIQ_ALLOW_MISSING_WEIGHTS", "0") == "1":
                warnings.warn(f"[VTAMIQ] pretrained=True but '{path}' does not exist: the transformer keeps its random "
                              "initialisation (VTAMIQ_ALLOW_MISSING_WEIGHTS=1)")
            else:                                  # np.load in the reference (transformer.py:622-624)
                raise FileNotFoundError(
                    f"[VTAMIQ] pretrained=True but the ViT checkpoint '{path}' does not exist.  Pass vit_config['vit_weights_path'], "
                    "set VTAMIQ_VIT_WEIGHTS, or vit_config['pretrained']=False when a full checkpoint is loaded afterwards "
                    "(VTAMIQ_ALLOW_MISSING_WEIGHTS=1 keeps the random initialisation with a warning)")

    # ---- reference surface --------------------------------------------------------------------------------
    @property
    def vit_hidden_size(self):                    # backbone.py:9-11
        return self.transformer.hidden_size

    @property
    def vit_num_layers(self):                     # backbone.py:13-15
        return len(self.transformer.encoder.layers)

    def set_freeze_state(self, freeze_state, freeze_dict):
        """requires_grad bookkeeping of vtamiq.py:81-92 / backbone.py:62-106 (host-side only; the engine never trains)."""
        requires_grad = not freeze_state
        fd = freeze_dict["freeze_dict_vit"]
        freeze_all = fd is None
        tr = self.transformer

        def set_grad(m, flag):
            for p in m.parameters():
                p.requires_grad = flag

        if freeze_all or fd["freeze_encoder"]:
            set_grad(tr.encoder, requires_grad)
            if not freeze_all and not fd["freeze_encoder_layerscale"] and tr.use_layer_scale:
                for layer in tr.encoder.layers:
                    set_grad(layer.ls1, True)
                    set_grad(layer.ls2, True)
        if freeze_all or fd["freeze_embeddings_cls_token"]:
            tr.embeddings.cls_token.requires_grad = requires_grad
        if (freeze_all or fd["freeze_embeddings_extra_tokens"]) and hasattr(tr.embeddings, "extra_tokens"):
            tr.embeddings.extra_tokens.requires_grad = requires_grad
        if (freeze_all or fd["freeze_embeddings_patch"]) and tr.embeddings.use_patch_embedding:
            set_grad(tr.embeddings.patch_embeddings, requires_grad)
        if (freeze_all or fd["freeze_embeddings_pos"]) and tr.embeddings.use_pos_embedding:
            set_grad(tr.embeddings.positional_embeddings, requires_grad)
        if (freeze_all or fd["freeze_embeddings_scale"]) and tr.embeddings.use_scale_embedding:
            set_grad(tr.embeddings.scale_embeddings, requires_grad)
        if freeze_dict["freeze_quality_decoder"]:
            set_grad(self.quality_decoder, requires_grad)
        if freeze_dict["freeze_q_predictor"]:
            set_grad(self.q_predictor, requires_grad)

    # ---- engine management --------------------------------------------------------------------------------
    def _engine_lib(self):
        """The C-ABI library this model's engine lives in: the product library (the fp8 experiment's subclass returns its own build)."""
        return _lib.load()

    def _check(self, rc: int) -> None:
        _lib.check(rc, self._engine_lib())

    def _walk(self):
        """The model's own tree as (module, name, parameter) and (parent, name, child) triples -- what _signature re-checks by identity
        on every forward.  No process-global torch hook is involved: other modules in the process cost nothing and
        change nothing here."""
        params, mods = [], []
        for mod in self.modules():
            for name, p in mod._parameters.items():
                if p is not None:
                    params.append((mod, name, p))
            for name, child in mod._modules.items():
                if child is not None:
                    mods.append((mod, name, child))
        return params, mods

    def _signature(self):
        """What the packed engine weights were made from: identity, storage and version of every parameter.  The cached walk of the
        module tree is re-validated by identity every forward (each parent still holds the same child module, each module the same
        Parameter object: ~350 `is` checks) -- which sees parameter or module REPLACEMENT and load_state_dict(assign=True) anywhere in
        THIS model's tree -- and re-done only when that fails; the rest is (data_ptr, _version) of the cached parameters, which sees
        `p.data = new_tensor` (new storage) and every in-place write that bumps the version counter (load_state_dict, optimizers).
        Not seen: in-place edits through `.data` (p.data.copy_()) -- refresh_weights()."""
        cache = self.__dict__.get("_param_cache")
        if cache is not None:
            params, mods = cache
            for mod, name, p in params:
                if mod._parameters.get(name) is not p:
                    cache = None
                    break
            else:
                for parent, name, child in mods:
                    if parent._modules.get(name) is not child:
                        cache = None
                        break
        if cache is None:
            cache = self._walk()
            self.__dict__["_param_cache"] = cache
        ps = [p for _, _, p in cache[0]]
        return (tuple(map(id, ps)), tuple(p.data_ptr() for p in ps), tuple(p._version for p in ps))

    def _release_engine(self, park: bool = False):
        """Destroy the engine -- or, park=True (the numerics mode changes on the same device: "auto" probing bf16x3 and coming back), keep
        it with its packed weights under its precision, so that a batch with non-finite inputs does not cost two engine builds and two
        weight re-packs per call.  Without park, parked engines are destroyed too."""
        eng = self.__dict__.get("_engine")
        parked = self.__dict__.setdefault("_parked", {})
        if eng is not None:
            if park:
                parked[self.__dict__.get("_engine_precision")] = (eng, self.__dict__.get("_weights_sig"))
            else:
                self._engine_lib().vtq_destroy(eng)
        if not park:
            for h, _ in parked.values():
                self._engine_lib().vtq_destroy(h)
            parked.clear()
        self.__dict__["_engine"] = None
        self.__dict__["_weights_sig"] = None

    def _forget_packed_weights(self):
        self.__dict__["_weights_sig"] = None
        parked = self.__dict__.get("_parked") or {}
        for k, (h, _) in list(parked.items()):
            parked[k] = (h, None)

    def __del__(self):
        try:
            self._release_engine()
        except Exception:           # interpreter shutdown
            pass

    def refresh_weights(self):
        """Force a re-pack of the parameters into the engine on the next forward.  REQUIRED after in-place edits through
        `.data` (p.data.copy_(), p.data.normal_(), ...): those do not bump the version counter `_signature` watches, so the
        engine would keep serving the previously packed weights.  load_state_dict / .to() / optimizer-style in-place ops on
        the parameters themselves are picked up automatically; weights.load_* call this for you."""
        self._forget_packed_weights()

    def check_inputs(self):
        """Synchronise and raise IndexError if any forward since the last check saw a position outside [0, 1), FloatingPointError
        if one produced a non-finite CLS difference (operand range overflow, fp16 modes)."""
        if self._engine is None:
            return
        flags = C.c_int32(0)
        with torch.cuda.device(self._engine_device):
            stream = torch.cuda.current_stream(self._engine_device).cuda_stream
            self._check(self._engine_lib().vtq_input_errors(self._engine, C.byref(flags), stream))
        flags.value |= self.__dict__.pop("_flags_seen", 0)       # bits an earlier look of a subclass already collected (and cleared)
        if flags.value & 1:
            raise IndexError("pos outside [0, 1): index out of range in the positional-embedding table (transformer.py:417-421)")
        if flags.value & 4:                         # raised by the fp8 experiment's kernels only
            raise FloatingPointError("fp8 experiment: an activation exceeded e4m3's range after scaling and was clamped to +-448 -- the "
                                     "calibrated scales do not fit this data; calibrate_fp8() on a representative batch")
        if flags.value & 2:
            raise FloatingPointError(
                f"non-finite encoder output in precision={self.engine_precision!r}: an activation or weight left the operand format's range "
                "(fp16 modes: |v| <= 65504) or the inputs held inf / NaN; precision='bf16x3' has the fp32 range")

    @property
    def engine_precision(self) -> str:
        """The numerics mode the engine runs: `precision`, or for "auto" fp16x3 until an operand overflow switched it to bf16x3."""
        if self.precision != "auto":
            return self.precision
        return AUTO_FALLBACK if self._auto_fallback else AUTO_FIRST

    def _read_flags(self) -> int:
        flags = C.c_int32(0)
        with torch.cuda.device(self._engine_device):
            stream = torch.cuda.current_stream(self._engine_device).cuda_stream
            self._check(self._engine_lib().vtq_input_errors(self._engine, C.byref(flags), stream))
        return flags.value

    def _enqueue(self, device, launch):
        """One forward on the engine of the current numerics mode (created / re-packed on demand)."""
        lib = self._ensure_engine(device)
        # vtamiq.py:57, 107-108: `token_num` picks the token row the head consumes ("can be CLS token or extra_token"); the reference
        # reads the attribute at every forward, so does this (a value outside the model's tokens is refused by the library)
        # (Python indexing into the (B, H, T) token rows: a negative index counts from the last register token)
        t = int(self.token_num)
        self._check(lib.vtq_set_iqa_token(self._engine, t + self.spec.num_tokens if t < 0 else t))
        launch(lib)

    def _launch_checked(self, device, launch):
        """Enqueue one forward (`launch(lib)`) and apply the model's input / range policy (see DEFAULT_PRECISION)."""
        self._enqueue(device, launch)
        if self.precision == "auto":
            flags = self._read_flags()
            if flags & 2 and not self._auto_fallback:
                # Either an operand left the fp16 range (the checkpoint's activation scale: bf16x3 cures it, and the switch stays), or
                # the inputs / weights hold inf / NaN (nothing cures that).  Run the call again in bf16x3 and look.
                self._auto_fallback = True
                self._enqueue(device, launch)
                flags2 = self._read_flags()
                if flags2 & 2:
                    # still non-finite with the fp32 operand range: the DATA is non-finite, not the format too narrow.  Go back to
                    # the parity mode (one bad batch must not leave the model in bf16x3 for good) and hand the caller
                    # what the reference's fp32 forward returns for such inputs: NaN scores (train.py:602-607), loudly.
                    self._auto_fallback = False
                    warnings.warn("[VTAMIQ] non-finite scores: the inputs or weights of this call hold inf / NaN (both the fp16x3 and "
                                  "the bf16x3 forward overflowed); the model stays in precision 'fp16x3'")
                    self._enqueue(device, launch)            # the scores handed back are the parity mode's (the healthy pairs' bits included)
                    flags2 = (flags2 & 1) | (self._read_flags() & 1)
                else:
                    # the switch is sticky: the parked fp16x3 engine (packed weights + workspace) will not be used again
                    for h, _ in self.__dict__.get("_parked", {}).values():
                        lib_ = self._engine_lib()
                        lib_.vtq_destroy(h)
                    self.__dict__.get("_parked", {}).clear()
                    warnings.warn(f"[VTAMIQ] an activation or weight left the fp16 operand range (|v| > 65504) in precision "
                                  f"{AUTO_FIRST!r}: this model now runs {AUTO_FALLBACK!r} (fp32 operand range, the same 3-MFMA "
                                  "split; the call was repeated)")
                flags = (flags & 1) | (flags2 & 1)
            elif flags & 2:
                warnings.warn("[VTAMIQ] non-finite scores in precision 'bf16x3': the inputs or weights of this call hold inf / NaN")
            if flags & 1:
                raise IndexError("pos outside [0, 1): index out of range in the positional-embedding table (transformer.py:417-421)")
        elif self.validate_inputs:
            self.check_inputs()

    def _ensure_engine(self, device: torch.device):
        lib = self._engine_lib()
        if self._engine is not None and self._engine_device != device:
            self._release_engine()
        elif self._engine is not None and self._engine_precision != self.engine_precision:
            self._release_engine(park=True)
        parked = self.__dict__.get("_parked") or {}
        if self._engine is None and self.engine_precision in parked:
            h, sig = parked.pop(self.engine_precision)
            self._engine, self._engine_device, self._engine_precision = h, device, self.engine_precision
            self._weights_sig = sig
        if self._engine is None:
            s = self.spec
            cfg = _lib.VtqConfig(
                hidden_size=s.hidden_size, mlp_dim=s.mlp_dim, num_heads=s.num_heads, num_layers=s.num_layers,
                patch_dim=s.patch_dim, pos_grid=s.pos_grid, num_extra_tokens=s.num_extra_tokens,
                num_scales=s.num_scales if s.use_scale_embedding else 0, use_layer_scale=int(s.use_layer_scale),
                calibrate=int(s.calibrate), diff_scale=int(s.diff_scale), num_rgs=s.num_rgs, num_rcabs=s.num_rcabs,
                ca_hidden=s.ca_hidden, precision=_PRECISIONS[self.engine_precision], num_adapters=s.num_adapters,
                options=self.engine_options)
            h = C.c_void_p()
            self._check(lib.vtq_create(C.byref(cfg), C.byref(h)))
            self._engine, self._engine_device, self._engine_precision = h, device, self.engine_precision
            self._weights_sig = None
            self._engine_created()
        sig = self._signature()
        if sig != self._weights_sig:
            sd = self.state_dict()
            names = [k for k, _, _ in self.spec.state_layout()]
            missing = [k for k in names if k not in sd]
            if missing:
                raise RuntimeError(f"state_dict is missing {missing[:3]}...")
            if not self.spec.use_pos_embedding:
                # use_pos_embedding=False (transformer.py:497-499, 514, 539): the model has no table and the reference adds nothing to the
                # patch rows or to CLS.  The engine's embedding epilogue always adds a table row: it gets a table of zeros (x + 0 = x).
                pk = "transformer.embeddings.positional_embeddings.positional_embeddings"
                names.append(pk)
                sd[pk] = torch.zeros(1, self.spec.pos_grid ** 2 + 1, self.spec.hidden_size, device=device, dtype=torch.float32)
            if not self.spec.use_patch_embedding:
                # use_patch_embedding=False (transformer.py:473-480): no Conv2d; the engine's patch GEMM is never launched for pre-embedded input
                # (vtq_forward_tokens), its weight slots just have to be filled
                sp = self.spec
                for pk, shape in (("transformer.embeddings.patch_embeddings.weight", (sp.hidden_size, 3, sp.patch_size, sp.patch_size)),
                                  ("transformer.embeddings.patch_embeddings.bias", (sp.hidden_size,))):
                    names.append(pk)
                    sd[pk] = torch.zeros(*shape, device=device, dtype=torch.float32)
            keep = []
            descs = (_lib.VtqTensorDesc * len(names))()
            for i, k in enumerate(names):
                t = sd[k].detach()
                if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.to(device=device, dtype=torch.float32).contiguous()
                keep.append(t)
                descs[i] = _lib.VtqTensorDesc(k.encode(), t.data_ptr(), t.numel())
            stream = torch.cuda.current_stream(device).cuda_stream
            reload_ = self._weights_sig is not None
            self._check(lib.vtq_load_weights(self._engine, descs, len(names), stream))
            self._weights_sig = sig
            self._weights_loaded(reload_)
        return lib

    def _engine_created(self):                    # hooks of the fp8 experiment's subclass (state that must follow the engine's lifetime)
        pass

    def _weights_loaded(self, reload_: bool):
        pass

    def _apply(self, fn, *a, **k):            # .to()/.cuda()/.float(): parameter storage (possibly the objects) is replaced
        self._forget_packed_weights()
        self.__dict__["_param_cache"] = None
        return super()._apply(fn, *a, **k)

    # ---- the hot path ---------------------------------------------------------------------------------------
    @staticmethod
    def _prep(t: torch.Tensor, device) -> torch.Tensor:
        if t.device != device:
            raise ValueError(f"all inputs must live on {device}, got {t.device}")
        if t.dtype != torch.float32:
            t = t.float()                     # the loader hands everything over as f32 (train.py:254-255)
        return t if t.is_contiguous() else t.contiguous()

    def forward(self, patches, pos, scales, _trace: Optional[torch.Tensor] = None):
        if self.training:
            raise NotImplementedError(
                "the MI355X engine implements the eval/no-grad forward only (Dropout/DropPath of vtamiq.py:72-75 and "
                "channel_attention.py:26-29 are train-time stochastic; backward is out of scope): call model.eval()")
        patches_ref, patches_dist = patches
        pos_ref, pos_dist = pos
        scales_ref, scales_dist = scales
        device = patches_ref.device
        if device.type != "cuda":
            raise RuntimeError("VTAMIQ (vtamiq_amd) runs on an MI355X only: move the model and inputs to 'cuda'. "
                               "There is no CPU fallback on the product path.")
        if torch.is_grad_enabled() and not self._warned_grad and any(p.requires_grad for p in self.parameters()):
            warnings.warn("vtamiq_amd.VTAMIQ.forward returns scores without an autograd graph (inference engine)")
            self._warned_grad = True
        # Embeddings.forward (transformer.py:527-535): a 5-D tensor goes through the patch convolution, anything else is taken as pre-embedded
        # (B, N, H) rows -- whatever `use_patch_embedding` says; a model built without the convolution fails on 5-D input like the reference does
        tokens_in = patches_ref.dim() != 5
        if tokens_in:
            if patches_ref.dim() != 3 or patches_ref.shape != patches_dist.shape or patches_ref.shape[2] != self.spec.hidden_size:
                raise ValueError(f"pre-embedded input must be two (B,N,{self.spec.hidden_size}) tensors, got {tuple(patches_ref.shape)} / {tuple(patches_dist.shape)}")
            B, N = patches_ref.shape[:2]
        else:
            if not self.spec.use_patch_embedding:
                raise AttributeError("'Embeddings' object has no attribute 'patch_embeddings' (use_patch_embedding=False: pass pre-embedded (B,N,H) rows)")
            if patches_ref.shape != patches_dist.shape:
                raise ValueError(f"patches must be two (B,N,3,P,P) tensors, got {tuple(patches_ref.shape)} / {tuple(patches_dist.shape)}")
            B, N, Cc, P, P2 = patches_ref.shape
            if (Cc, P, P2) != (3, self.spec.patch_size, self.spec.patch_size):
                raise ValueError(f"patch shape {(Cc, P, P2)} != (3,{self.spec.patch_size},{self.spec.patch_size})")
        if not self.spec.use_pos_embedding:
            # the reference never looks at `pos` then (transformer.py:539): whatever was passed, None included, has no effect.  The
            # engine's index kernel still wants coordinates: zeros (table row 1 of the all-zero table _ensure_engine installs)
            pos_ref = pos_dist = torch.zeros(B, N, 2, device=device, dtype=torch.float32)
        if tuple(pos_ref.shape) != (B, N, 2) or tuple(pos_dist.shape) != (B, N, 2):
            raise ValueError("pos must be two (B,N,2) tensors")
        use_scales = self.spec.use_scale_embedding
        if use_scales:
            if scales_ref is None or scales_dist is None:
                raise ValueError("Model uses scale embedding but scales is passed as None.")   # transformer.py:547-548
            if scales_ref.numel() != B * N or scales_dist.numel() != B * N:
                raise ValueError("scales must be two (B,N) tensors")
        with torch.cuda.device(device):
            pr, pd = self._prep(patches_ref, device), self._prep(patches_dist, device)
            qr, qd = self._prep(pos_ref, device), self._prep(pos_dist, device)
            sr = self._prep(scales_ref, device) if use_scales else None
            sdist = self._prep(scales_dist, device) if use_scales else None
            q = torch.empty(B, device=device, dtype=torch.float32)
            stream = torch.cuda.current_stream(device).cuda_stream

            def launch(lib):
                if _trace is not None:
                    self._check(lib.vtq_set_token_trace(self._engine, _trace.data_ptr()))
                try:
                    self._check((lib.vtq_forward_tokens if tokens_in else lib.vtq_forward)(self._engine, pr.data_ptr(), pd.data_ptr(), qr.data_ptr(), qd.data_ptr(),
                                               sr.data_ptr() if use_scales else None, sdist.data_ptr() if use_scales else None,
                                               B, N, q.data_ptr(), stream))
                finally:
                    if _trace is not None:
                        lib.vtq_set_token_trace(self._engine, None)
            self._launch_checked(device, launch)
        return q, None

    def forward_pairwise(self, patches, pos, scales):
        """Pairwise items: `patches = (p_ref, p_dist1, p_dist2)` etc.  Returns (q1, q2) == what two calls
        `self((p_ref, p_dist1), ...)[0]`, `self((p_ref, p_dist2), ...)[0]` give (train.py:286-287), bit for bit, with the
        reference image encoded once: 3B sequences instead of 4B."""
        if self.training:
            raise NotImplementedError("the MI355X engine implements the eval/no-grad forward only: call model.eval()")
        if len(patches) != 3 or (self.spec.use_pos_embedding and len(pos) != 3):
            raise ValueError("forward_pairwise expects (ref, dist1, dist2) triplets")
        device = patches[0].device
        if device.type != "cuda":
            raise RuntimeError("VTAMIQ (vtamiq_amd) runs on an MI355X only: move the model and inputs to 'cuda'. "
                               "There is no CPU fallback on the product path.")
        B, N = patches[0].shape[:2]
        # Embeddings.forward (transformer.py:527-535), as in forward(): 5-D tensors go through the patch convolution, 3-D ones are pre-embedded rows
        tokens_in = patches[0].dim() != 5
        for t in patches:
            if tokens_in:
                if tuple(t.shape) != (B, N, self.spec.hidden_size):
                    raise ValueError(f"pre-embedded input must be three (B,N,{self.spec.hidden_size}) tensors, got {tuple(t.shape)}")
            elif not self.spec.use_patch_embedding:
                raise AttributeError("'Embeddings' object has no attribute 'patch_embeddings' (use_patch_embedding=False: pass pre-embedded (B,N,H) rows)")
            elif tuple(t.shape) != (B, N, 3, self.spec.patch_size, self.spec.patch_size):
                raise ValueError(f"patches must be three (B,N,3,P,P) tensors, got {tuple(t.shape)}")
        if not self.spec.use_pos_embedding:            # as in forward(): `pos` is not looked at
            pos = (torch.zeros(B, N, 2, device=device, dtype=torch.float32),) * 3
        for t in pos:
            if tuple(t.shape) != (B, N, 2):
                raise ValueError("pos must be three (B,N,2) tensors")
        use_scales = self.spec.use_scale_embedding
        if use_scales and (scales is None or any(t is None for t in scales)):
            raise ValueError("Model uses scale embedding but scales is passed as None.")
        with torch.cuda.device(device):
            pt = [self._prep(t, device) for t in patches]
            ps = [self._prep(t, device) for t in pos]
            sc = [self._prep(t, device) for t in scales] if use_scales else None
            q = torch.empty(2 * B, device=device, dtype=torch.float32)
            arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
            stream = torch.cuda.current_stream(device).cuda_stream
            self._launch_checked(device, lambda lib: self._check((lib.vtq_forward_pairwise_tokens if tokens_in else lib.vtq_forward_pairwise)(
                self._engine, arr(pt), arr(ps), arr(sc) if use_scales else None, B, N, q.data_ptr(), stream)))
        return q[:B], q[B:]

    # ---- measurement helpers (bench.py) ---------------------------------------------------------------------
    def profile_enable(self, classes):
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES.index(c)
        self._check(self._engine_lib().vtq_profile_enable(self._engine, mask))

    def profile_collect(self):
        n = len(_lib.KERNEL_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self._check(self._engine_lib().vtq_profile_collect(self._engine, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(_lib.KERNEL_CLASSES)}

    def workspace_bytes(self, B, N):
        return int(self._engine_lib().vtq_workspace_bytes(self._engine, B, N)) if self._engine is not None else 0

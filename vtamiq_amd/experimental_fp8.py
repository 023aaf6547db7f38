"""The fp8 EXPERIMENT (BASELINE.json configs[4]) -- not part of the product path.

`VTAMIQFp8` is `vtamiq_amd.VTAMIQ` plus the e4m3 mode: linear layers on OCP e4m3 operands with the MX-scaled MFMA (2x the bf16 MFMA rate),
weights with per-output-channel power-of-two scales, activations with per-tensor scales calibrated on the first batch.  It lives in its own build of the
engine sources (include/vtamiq_hip_fp8.h; `python -m vtamiq_amd.build --fp8` -> vtamiq_amd/libvtamiq_hip_fp8.so, which `__graft_entry__.build()`
makes next to the product library) and reaches it through its own ctypes handle (`_lib.load_fp8()`): no environment variable, and the
product library is still the one every other model in the process uses.

Why it is here and not in `VTAMIQ`: it is a THROUGHPUT experiment, not a scoring mode -- 3 mantissa bits of activation
precision put its scores tens of percent from the fp32 model's on random-init weights and SROCC 0.66 - 0.84 from them on a distortion ladder
with every scale granularity tried (per-tensor, per-row, MX blocks: profiles/r04_fp8_study.txt); its parity statement is against its own
fake-quant oracle (oracle/fp8_oracle.py), which nothing from the reference pins.

Scale ownership: scales the USER installed (`set_fp8_scales`, `broadcast_fp8_scales`) belong to the model and survive weight
reloads and engine re-creation; AUTO-calibrated scales belong to one (engine, weights) pair -- whenever weights are (re)loaded into an engine
they are dropped on both sides of the ABI (`vtq_fp8_reset`) and the next forward calibrates again.
"""
from __future__ import annotations

import ctypes as C
import warnings

import torch
import torch.distributed as dist

from . import _lib
from .model import VTAMIQ


class VTAMIQFp8(VTAMIQ):
    _FP8_EXPERIMENT = True

    def __init__(self, *args, precision: str = "fp8", **kwargs):
        _lib.load_fp8()                                   # RuntimeError when the experiment's library is not built
        super().__init__(*args, precision=precision, **kwargs)
        if self.precision == "fp8" and self.spec.num_adapters > 0:
            raise NotImplementedError("adapters are not available in the fp8 mode (the adapter input would need an e4m3 copy of the "
                                      "branch output); use a 16-bit precision")

    def _engine_lib(self):
        return _lib.load_fp8()

    # ---- state that follows the engine's lifetime -------------------------------------------------------------------------------
    def _weights_loaded(self, reload_: bool):
        if self.engine_precision != "fp8":
            return
        d = self.__dict__
        if d.get("_fp8_user") and d.get("_fp8_saved") is not None:
            self._install_fp8(d["_fp8_saved"])            # the user's scales: the model's, whatever engine or weights
            return
        # auto-calibrated scales fitted the weights the engine held before: forget them on both sides, the next forward calibrates
        d["_fp8_saved"] = None
        d["_fp8_checked"] = 0
        self._check(self._engine_lib().vtq_fp8_reset(self._engine))

    def _launch_checked(self, device, launch):
        if self.engine_precision != "fp8":
            return super()._launch_checked(device, launch)
        self._enqueue(device, launch)
        d = self.__dict__
        # the first forward after a (re)load calibrated the activation scales on its batch: remember them, and look at the saturation
        # bit on the first few forwards -- an asynchronous fp8 model must not clamp at +-448 unnoticed
        if d.get("_fp8_saved") is None:
            self.fp8_scales()
        n = d.get("_fp8_checked", 0)
        if self.validate_inputs:
            self.check_inputs()                           # the documented contract: IndexError / FloatingPointError after every forward
        elif n < 3:
            d["_fp8_checked"] = n + 1
            flags = self._read_flags()
            d["_flags_seen"] = d.get("_flags_seen", 0) | flags          # check_inputs() still reports them
            if flags & 4:
                warnings.warn("[VTAMIQ] fp8 experiment: an activation exceeded e4m3's range after scaling and was clamped to +-448 -- the "
                              "activation scales do not fit this data; calibrate_fp8() on a representative batch")

    # ---- activation scales ------------------------------------------------------------------------------------------------------
    def fp8_scales(self):
        """The engine's per-tensor activation scales: {"patch": s, "ln1": [L], "att": [L], "ln2": [L], "gelu": [L]} (powers of two).
        They are calibrated on the batch of the first forward after every weight load (include/vtamiq_hip_fp8.h); calibrate_fp8()
        repeats that on a batch of your choice, set_fp8_scales() installs given ones."""
        if self._engine is None or self.engine_precision != "fp8":
            raise RuntimeError("fp8_scales: no fp8 engine yet (run a forward first)")
        lib = self._engine_lib()
        n = lib.vtq_fp8_get_scales(self._engine, None, 0)
        buf = (C.c_float * n)()
        lib.vtq_fp8_get_scales(self._engine, buf, n)
        v = list(buf)
        L = (n - 1) // 4
        sc = {"patch": v[0], "ln1": v[1::4][:L], "att": v[2::4][:L], "ln2": v[3::4][:L], "gelu": v[4::4][:L]}
        if not self.__dict__.get("_fp8_user"):
            self.__dict__["_fp8_saved"] = sc
        return sc

    def _install_fp8(self, sc):
        L = len(sc["ln1"])
        flat = [sc["patch"]]
        for i in range(L):
            flat += [sc["ln1"][i], sc["att"][i], sc["ln2"][i], sc["gelu"][i]]
        buf = (C.c_float * len(flat))(*flat)
        self._check(self._engine_lib().vtq_fp8_set_scales(self._engine, buf, len(flat)))

    def set_fp8_scales(self, sc):
        """Install activation scales (the dict fp8_scales() returns, e.g. from a checkpoint's side file or from rank 0:
        broadcast_fp8_scales).  Installed scales survive weight reloads and engine re-creation; calibrate_fp8() replaces them."""
        if self.precision != "fp8":
            raise RuntimeError("set_fp8_scales: precision is not 'fp8'")
        self.__dict__["_fp8_saved"] = {k: (list(v) if isinstance(v, (list, tuple)) else float(v)) for k, v in sc.items()}
        self.__dict__["_fp8_user"] = True
        if self._engine is not None:
            self._install_fp8(self.__dict__["_fp8_saved"])

    def calibrate_fp8(self, patches, pos, scales):
        """Re-calibrate the activation scales on this batch (arguments as forward()); returns the batch's scores."""
        if self.precision != "fp8":
            raise RuntimeError("calibrate_fp8: precision is not 'fp8'")
        device = patches[0].device
        use_scales = self.spec.use_scale_embedding
        with torch.cuda.device(device):
            self.__dict__["_fp8_user"] = False            # before the engine is (re)built: its weight load must not re-install old user scales
            lib = self._ensure_engine(device)
            t = [self._prep(x, device) for x in (patches[0], patches[1], pos[0], pos[1])]
            sc = [self._prep(x, device) for x in scales] if use_scales else [None, None]
            B, N = patches[0].shape[:2]
            q = torch.empty(B, device=device, dtype=torch.float32)
            stream = torch.cuda.current_stream(device).cuda_stream
            self._check(lib.vtq_fp8_calibrate(self._engine, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                             sc[0].data_ptr() if use_scales else None, sc[1].data_ptr() if use_scales else None,
                                             B, N, q.data_ptr(), stream))
        self.__dict__["_fp8_checked"] = 0
        self.fp8_scales()
        return q


def broadcast_fp8_scales(model: VTAMIQFp8, src: int = 0, group=None) -> None:
    """Every rank calibrates its activation scales on its OWN shard's first batch, so the ranks of a data-parallel job would score with
    different scales.  Call this once after the first forward (or after calibrate_fp8 on the source rank): that rank's scales are installed
    on every rank (model.set_fp8_scales), which also keeps them across weight reloads.  `src` is a GLOBAL rank, as
    torch.distributed.broadcast_object_list takes it (a group-local comparison picked the wrong rank in a subgroup)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    box = [model.fp8_scales() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    model.set_fp8_scales(box[0])


def model_class(precision: str):
    """`VTAMIQFp8` for precision "fp8", the product `VTAMIQ` for every other mode (measurement scripts that loop over modes)."""
    return VTAMIQFp8 if precision in ("fp8", "fp8w") else VTAMIQ

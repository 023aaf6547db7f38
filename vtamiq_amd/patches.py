"""On-device image -> patch tensor: uint8 images + host-sampled coordinates in, the model's
(patches, pos, scales) call tensors out -- the GPU-side half of the reference's loader item
(data/patch_datasets.py:397-409: transform_img x K, get_iqa_patches).  Coordinates stay CPU-sampled (RNG parity is not required)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib


def check_samples_host(samples: torch.Tensor, scale_ids: Optional[torch.Tensor], H: int, W: int, num_scales: int = 1, patch_size: int = 16) -> None:
    """The range check of the sampled coordinates (IndexError like the reference's numpy fancy-indexing).  Works on the tensors where they
    are: host tensors cost no GPU synchronisation (a pipelined loader calls this BEFORE the copy and passes validate=False below)."""
    P = int(patch_size)
    if samples.device.type == "cpu" and (scale_ids is None or scale_ids.device.type == "cpu"):
        # numpy on the calling thread: torch CPU operators on these small tensors wake the intra-op thread pool, whose spinning workers
        # then compete with the HIP runtime's own threads -- measured: a pipelined loop fell from 15.4 to 31 ms per batch with the
        # torch form of this check on the host (tools/e2e_probe.py)
        import numpy as np
        smp_np = samples.detach().numpy()
        lvl_np = scale_ids.detach().numpy().astype(np.int64) if scale_ids is not None else np.zeros(smp_np.shape[:2], dtype=np.int64)
        if ((lvl_np < 0) | (lvl_np >= num_scales)).any():
            raise IndexError("scale_ids outside [0, num_scales)")
        hmax_np = np.array([(H >> s) - P for s in range(num_scales)])[lvl_np]
        wmax_np = np.array([(W >> s) - P for s in range(num_scales)])[lvl_np]
        if ((smp_np[..., 0] < 0) | (smp_np[..., 0] > hmax_np) | (smp_np[..., 1] < 0) | (smp_np[..., 1] > wmax_np)).any():
            raise IndexError(f"patch sample outside its pyramid level (row in [0, h-{P}], col in [0, w-{P}] required)")
        return
    smp = samples.to(torch.int64)
    lvl = scale_ids.to(device=smp.device, dtype=torch.int64) if scale_ids is not None else torch.zeros(smp.shape[:2], dtype=torch.int64, device=smp.device)
    if bool(((lvl < 0) | (lvl >= num_scales)).any()):
        raise IndexError("scale_ids outside [0, num_scales)")
    hmax = torch.tensor([(H >> s) - P for s in range(num_scales)], device=smp.device)[lvl]
    wmax = torch.tensor([(W >> s) - P for s in range(num_scales)], device=smp.device)[lvl]
    if bool(((smp[..., 0] < 0) | (smp[..., 0] > hmax) | (smp[..., 1] < 0) | (smp[..., 1] > wmax)).any()):
        raise IndexError(f"patch sample outside its pyramid level (row in [0, h-{P}], col in [0, w-{P}] required)")


def extract_patches(images_u8: torch.Tensor, samples: torch.Tensor, scale_ids: Optional[torch.Tensor] = None, num_scales: int = 1,
                    flips: Optional[torch.Tensor] = None, mean: Sequence[float] = (0.5, 0.5, 0.5), std: Sequence[float] = (0.5, 0.5, 0.5),
                    patch_size: int = 16, validate: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """images_u8 [NI, H, W, 3] uint8 (cuda); samples [NI, N, 2] int32 (row, col at the patch's own scale);
    scale_ids [NI, N] int32 (required when num_scales > 1, patches of scale s index pyramid level s); flips [NI, 2] int32
    (hflip, vflip) or None; patch_size P = 16 (ViT-B16 / ViT-L16) or 8 (ViT-B8).  Returns patches [NI, N, 3, P, P] f32, pos [NI, N, 2] f32,
    scales [NI, N] f32 or None.  validate=False skips the device-side range check of the samples (one reduction + a stream
    synchronisation): for a pipelined loader that has checked them on the host before the copy (check_samples_host)."""
    lib = _lib.load()
    dev = images_u8.device
    if dev.type != "cuda":
        raise RuntimeError("extract_patches runs on the GPU only (no CPU fallback on the product path)")
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise ValueError("images_u8 must be uint8 [NI, H, W, 3]")
    NI, H, W, _ = images_u8.shape
    N = samples.shape[1]
    if tuple(samples.shape) != (NI, N, 2):
        raise ValueError("samples must be int32 [NI, N, 2]")
    if num_scales > 1 and scale_ids is None:
        raise ValueError("scale_ids are required when num_scales > 1")
    if not 1 <= num_scales <= 4:
        raise ValueError("1 <= num_scales <= 4")
    P = int(patch_size)
    if P not in (16, 8):
        raise ValueError("patch_size must be 16 or 8")
    # Range checks the reference gets for free from numpy fancy-indexing (an out-of-range patch raises IndexError there); the
    # gather kernel itself does not bounds-check.  One small reduction + sync per call, on the loader side of the pipeline.
    if (H >> (num_scales - 1)) < P or (W >> (num_scales - 1)) < P:
        raise ValueError(f"pyramid level {num_scales - 1} of a {H}x{W} image is smaller than one {P}x{P} patch")
    if scale_ids is not None and tuple(scale_ids.shape) != (NI, N):
        raise ValueError("scale_ids must be int32 [NI, N]")
    if validate:
        check_samples_host(samples, scale_ids, H, W, num_scales, P)
    images_u8 = images_u8.contiguous()
    samples = samples.to(device=dev, dtype=torch.int32).contiguous()
    sid = scale_ids.to(device=dev, dtype=torch.int32).contiguous() if scale_ids is not None else None
    fl = flips.to(device=dev, dtype=torch.int32).contiguous() if flips is not None else None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        levels = [torch.empty(NI, 3, H, W, device=dev, dtype=torch.float32)]
        m = (C.c_float * 3)(*mean)
        s = (C.c_float * 3)(*std)
        _lib.check(lib.vtq_k_image_normalize(images_u8.data_ptr(), levels[0].data_ptr(), NI, H, W, fl.data_ptr() if fl is not None else None,
                                             m, s, stream))
        for _ in range(1, num_scales):
            h, w = levels[-1].shape[2:]
            nxt = torch.empty(NI, 3, h // 2, w // 2, device=dev, dtype=torch.float32)
            _lib.check(lib.vtq_k_avgpool2(levels[-1].data_ptr(), nxt.data_ptr(), NI * 3, h, w, stream))
            levels.append(nxt)
        patches = torch.empty(NI, N, 3, P, P, device=dev, dtype=torch.float32)
        pos = torch.empty(NI, N, 2, device=dev, dtype=torch.float32)
        scales = torch.empty(NI, N, device=dev, dtype=torch.float32) if num_scales > 1 else None
        ptrs = (C.c_void_p * len(levels))(*[l.data_ptr() for l in levels])
        hs = (C.c_int32 * len(levels))(*[l.shape[2] for l in levels])
        ws = (C.c_int32 * len(levels))(*[l.shape[3] for l in levels])
        _lib.check(lib.vtq_k_gather_patches(ptrs, hs, ws, len(levels), samples.data_ptr(), sid.data_ptr() if sid is not None else None,
                                            patches.data_ptr(), pos.data_ptr(), scales.data_ptr() if scales is not None else None, NI, N,
                                            P, stream))
    return patches, pos, scales

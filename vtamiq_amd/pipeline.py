"""Host images in, scores out: the loader side of a validation / test pass, pipelined (SURVEY.md 8f-1 + 8f-4).

The reference's loader builds the fp32 patch tensor on the CPU (data/patch_datasets.py:397-409, data/patch_sampling.py:529-611) and ships
3.08 MB per pair to the GPU; here the CPU only SAMPLES patch coordinates (RNG parity is not required) and the uint8 images themselves cross
PCIe (1.19 MB per pair at 384 x 512): pinned host buffers -> H2D on a copy stream -> patches.extract_patches (normalise, pyramid, gather on
the GPU) -> the model's forward, with `depth` buffer sets so that the copy of batch i + 1 runs under the forward of batch i.  Scores stay on
the device (validate.compute_correlations_cat_flat reduces them there).

    pipe = ImagePairPipeline(model, pairs_per_batch=32, image_hw=(384, 512), patches=500)
    for i, (ref_u8, dist_u8, samples) in enumerate(loader):          # numpy / torch uint8 [B, H, W, 3] each, int32 [B, N, 2] (aligned) or [2B, N, 2]
        q = pipe.submit(ref_u8, dist_u8, samples)                     # device tensor [B]; returns as soon as the work is enqueued
        scores.append(q)
    # or, without the host copy: img, smp, sid = pipe.acquire(); <decode into the pinned views>; q = pipe.launch()

Measured (bench.py `e2e`, profiles/r05_bench_line.json): the loop sustains 0.98 x the forward's own throughput at B = 32, N = 500; with a 1 280-pair
validation set's final reductions (rank statistics + the logistic fit, once per set) 0.91 - 0.95 x by box.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .patches import check_samples_host, extract_patches


class ImagePairPipeline:
    def __init__(self, model, pairs_per_batch: int, image_hw: Tuple[int, int], patches: int, num_scales: int = 1,
                 device: Optional[torch.device] = None, depth: int = 2, mean: Sequence[float] = (0.5, 0.5, 0.5),
                 std: Sequence[float] = (0.5, 0.5, 0.5)):
        if depth < 2:
            raise ValueError("depth >= 2 (one buffer set is being filled while another is consumed)")
        self.model = model
        self.B, self.N, self.num_scales = int(pairs_per_batch), int(patches), int(num_scales)
        self.H, self.W = int(image_hw[0]), int(image_hw[1])
        self.P = model.spec.patch_size
        self.mean, self.std = tuple(mean), tuple(std)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("ImagePairPipeline runs on the GPU only (no CPU fallback on the product path)")
        NI = 2 * self.B
        self.depth = depth
        self.host_img = [torch.empty(NI, self.H, self.W, 3, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.host_smp = [torch.empty(NI, self.N, 2, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.host_sid = [torch.zeros(NI, self.N, dtype=torch.int32).pin_memory() for _ in range(depth)] if num_scales > 1 else None
        self.dev_img = [torch.empty(NI, self.H, self.W, 3, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.dev_smp = [torch.empty(NI, self.N, 2, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.dev_sid = [torch.empty(NI, self.N, dtype=torch.int32, device=self.device) for _ in range(depth)] if num_scales > 1 else None
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.consumed = [torch.cuda.Event() for _ in range(depth)]
        main = torch.cuda.current_stream(self.device)
        for ev in self.consumed:
            ev.record(main)
        self.count = 0

    @staticmethod
    def _np(a):
        return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)

    def acquire(self):
        """The next slot's PINNED host buffers as numpy views -- images uint8 [2B, H, W, 3] (ref rows first), samples int32 [2B, N, 2], scale ids
        int32 [2B, N] or None -- once the batch that used them `depth` submissions ago no longer reads them.  A loader that decodes straight
        into these views saves the host copy of submit(); follow with launch()."""
        s = self.count % self.depth
        self.consumed[s].synchronize()                       # the slot's buffers are no longer being copied from / gathered from (batch i - depth)
        return self.host_img[s].numpy(), self.host_smp[s].numpy(), (self.host_sid[s].numpy() if self.host_sid is not None else None)

    def launch(self) -> torch.Tensor:
        """Enqueue the slot filled through acquire(): range check of the samples on the host, H2D on the copy stream, gather + forward on the
        current stream.  Returns the batch's scores as a device tensor [B]."""
        s = self.count % self.depth
        self.count += 1
        B = self.B
        # the range check the reference gets from numpy fancy-indexing, on the host copy (no GPU synchronisation, no torch CPU operators)
        check_samples_host(self.host_smp[s], self.host_sid[s] if self.host_sid is not None else None, self.H, self.W, self.num_scales, self.P)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[s])    # the slot's device buffers were read by batch i - depth
            self.dev_img[s].copy_(self.host_img[s], non_blocking=True)
            self.dev_smp[s].copy_(self.host_smp[s], non_blocking=True)
            if self.dev_sid is not None:
                self.dev_sid[s].copy_(self.host_sid[s], non_blocking=True)
            self.copied[s].record(self.copy_stream)
        main.wait_event(self.copied[s])
        patches, pos, scales = extract_patches(self.dev_img[s], self.dev_smp[s], self.dev_sid[s] if self.dev_sid is not None else None,
                                               self.num_scales, mean=self.mean, std=self.std, patch_size=self.P, validate=False)
        self.consumed[s].record(main)                        # behind the gather: the uint8 images and the samples of the slot are free again
        sc = (scales[:B], scales[B:]) if scales is not None else (None, None)
        with torch.no_grad():
            return self.model((patches[:B], patches[B:]), (pos[:B], pos[B:]), sc)[0]

    def submit(self, ref_u8, dist_u8, samples, scale_ids=None) -> torch.Tensor:
        """One batch of B pairs: ref / dist uint8 images [B, H, W, 3] (host), sampled patch coordinates int32 [B, N, 2] (shared by ref and
        dist: aligned sampling, the reference's default) or [2B, N, 2] (ref rows first), scale ids [B, N] / [2B, N] when num_scales > 1.
        Returns the batch's scores as a device tensor [B]; everything is enqueued asynchronously."""
        B = self.B
        img, hs, hd = self.acquire()
        img[:B], img[B:] = self._np(ref_u8), self._np(dist_u8)
        smp = self._np(samples)
        if smp.shape[0] == B:
            hs[:B], hs[B:] = smp, smp
        else:
            hs[:] = smp
        if hd is not None:
            if scale_ids is None:
                raise ValueError("scale_ids are required when num_scales > 1")
            sid = self._np(scale_ids)
            if sid.shape[0] == B:
                hd[:B], hd[B:] = sid, sid
            else:
                hd[:] = sid
        return self.launch()

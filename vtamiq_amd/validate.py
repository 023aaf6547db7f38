"""Validation loop with the reductions on the device.

Host-side mirror of the reference's test-time loop for the accelerated path:
  * do_validation                   train.py:583-644   (no_grad / eval, `num_repeats` passes over the loader, scores
                                                         concatenated pass-major, correlations at the end)
  * average_over_repeats            train.py:398-400
  * compute_correlations_cat_flat   train.py:403-409
  * compute_correlations            utils/misc/correlations.py:21-52

What changes against the reference: scores stay on the GPU for the whole loop (the reference synchronises on `q.cpu()` after
every batch, train.py:617-618), the mean over repeats, normalisation, ranks, Kendall pair counts, Pearson and RMSE run as
HIP kernels (csrc/metrics.hip) in fp64, and only the logistic fit of PLCC / RMSE (scipy.optimize.leastsq on N numbers, as in
the reference) is host work.  Losses / tensorboard logging (train.py:606-625) belong to the training plane and are not
mirrored.  `predict_repeats` additionally runs the R repeats of one batch as a single R*B-pair forward.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .predict import get_data_tuple, predict

SROCC_FIELD, KROCC_FIELD, PLCC_FIELD, RMSE_FIELD = "SROCC", "KROCC", "PLCC", "RMSE"
PLCC_NOFIT_FIELD, RMSE_NOFIT_FIELD = "PLCC_NOFIT", "RMSE_NOFIT"
CORRELATIONS_EPS = 1e-6


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: device tensors only (there is no CPU fallback on this path)")


def average_over_repeats(x: torch.Tensor, num_repeats: int) -> torch.Tensor:
    """x: device tensor of N*num_repeats scores, pass-major -> fp64 [N] mean over the repeats (train.py:398-400)."""
    _need_cuda(x, "average_over_repeats")
    x = x.detach().reshape(-1).float().contiguous()
    if num_repeats < 1 or x.numel() % num_repeats:
        raise ValueError(f"cannot reshape array of size {x.numel()} into shape ({num_repeats}, -1)")
    n = x.numel() // num_repeats
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    _lib.check(_lib.load().vtq_k_repeat_mean(x.data_ptr(), out.data_ptr(), num_repeats, n, _stream()))
    return out


def _fit_function1(p, x):
    """correlations.py:124-127."""
    return p[0] * (0.5 - 1.0 / (1.0 + np.exp(p[1] * (x - p[2]) + CORRELATIONS_EPS))) + abs(p[3]) * x + p[4]


def _enqueue_rank_metrics(a: torch.Tensor, b: torch.Tensor, normalize: bool):
    """Device part: ranks / Kendall pair counts / Pearson / RMSE kernels and ONE device -> host copy into pinned memory, all enqueued on the
    current stream; -> (pinned host vector, n, event recorded behind the copy).  Nothing here waits for the GPU."""
    _need_cuda(a, "compute_correlations")
    _need_cuda(b, "compute_correlations")
    a = a.detach().reshape(-1).double().contiguous()
    b = b.detach().reshape(-1).double().contiguous()
    n = a.numel()
    if b.numel() != n or n < 2:
        raise ValueError("compute_correlations: need two vectors of equal length >= 2")
    work = torch.empty(4 * n, dtype=torch.float64, device=a.device)
    counts = torch.empty(3, dtype=torch.int64, device=a.device)
    out = torch.empty(3, dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().vtq_k_rank_metrics(a.data_ptr(), b.data_ptr(), n, int(bool(normalize)), work.data_ptr(),
                                              counts.data_ptr(), out.data_ptr(), _stream()))
    dev = torch.cat([out, counts.double(), work[:2 * n]])
    host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
    host.copy_(dev, non_blocking=True)                                # the loop's one device -> host copy
    ev = torch.cuda.Event()
    ev.record()
    return host, n, ev, (a, b, work, dev)                             # the device tensors stay referenced until the copy has run


def _finish_correlations(host: np.ndarray, n: int) -> dict:
    """Host part: Kendall's tau-b from the exact pair counts, the logistic fit (scipy.optimize.leastsq, as the reference), PLCC / RMSE."""
    import scipy.optimize
    spearman, pearson_nofit, rmse_nofit = float(host[0]), float(host[1]), float(host[2])
    con_minus_dis, xtie, ytie = (int(round(v)) // 2 for v in host[3:6])
    aa, bb = host[6:6 + n], host[6 + n:6 + 2 * n]
    tot = n * (n - 1) // 2
    if xtie == tot or ytie == tot:                                    # scipy.stats.kendalltau: a constant input has no tau
        kendall = float("nan")
    else:
        kendall = min(1.0, max(-1.0, con_minus_dis / np.sqrt(tot - xtie) / np.sqrt(tot - ytie)))       # tau-b
    fitted = bb
    try:                                                              # correlations.py:35-40 (FitFunction form 1, 'L1' residuals)
        guess = (1.0, 1.0, np.median(bb), 1.0, np.median(aa))
        p = scipy.optimize.leastsq(lambda q, x, y: y - _fit_function1(q, x), guess, args=(bb, aa), full_output=True)[0]
        if np.isnan(np.asarray(p)).any():
            raise OverflowError("Fitting failed: result contains NaNs.")
        fitted = _fit_function1(p, bb)
    except OverflowError:
        pass
    xm, ym = aa - aa.mean(), fitted - fitted.mean()
    with np.errstate(divide="ignore", invalid="ignore"):              # a constant input has no correlation: nan, as scipy.stats.pearsonr
        pearson = float(np.clip(np.dot(xm / np.linalg.norm(xm), ym / np.linalg.norm(ym)), -1.0, 1.0))
    rmse = float(np.sqrt(np.mean((aa - fitted) ** 2)))
    return {SROCC_FIELD: spearman, KROCC_FIELD: kendall, PLCC_FIELD: pearson, RMSE_FIELD: rmse,
            PLCC_NOFIT_FIELD: pearson_nofit, RMSE_NOFIT_FIELD: rmse_nofit}


def compute_correlations(a: torch.Tensor, b: torch.Tensor, normalize: bool = True) -> dict:
    """a (targets), b (predictions): device vectors.  Same six fields as utils/misc/correlations.py:21-52."""
    host, n, ev, keep = _enqueue_rank_metrics(a, b, normalize)
    ev.synchronize()
    return _finish_correlations(host.numpy(), n)


class PendingCorrelations:
    """The correlations of a validation set whose host part (the logistic fit: ~1 000 residual evaluations by MINPACK, 30 - 40 ms for 1 280
    scores) runs on a worker thread: the caller goes on enqueueing the NEXT pass (train.py runs a validation and a test pass per epoch,
    train.py:583-644) and asks for `result()` when it needs the numbers.  The device part was enqueued by the constructor's caller; the
    worker waits for ITS event, never for the device."""
    _pool = None

    def __init__(self, host, n, ev, keep):
        import concurrent.futures
        if PendingCorrelations._pool is None:
            PendingCorrelations._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="vtq-fit")
        self._keep = keep

        def work():
            ev.synchronize()
            return _finish_correlations(host.numpy(), n)
        self._future = PendingCorrelations._pool.submit(work)

    def done(self) -> bool:
        return self._future.done()

    def result(self) -> dict:
        r = self._future.result()
        self._keep = None
        return r


def compute_correlations_deferred(a: torch.Tensor, b: torch.Tensor, normalize: bool = True) -> PendingCorrelations:
    """compute_correlations without waiting: device reductions + the copy are enqueued, the host's fit runs on a worker thread."""
    return PendingCorrelations(*_enqueue_rank_metrics(a, b, normalize))


def compute_correlations_cat_flat(ys, yp, num_repeats: int = 1, defer: bool = False):
    """ys, yp: lists of per-batch device tensors in loop order (train.py:403-409).  defer=True: a PendingCorrelations (the fit on a worker thread)."""
    y = torch.cat([t.detach().reshape(-1).float() for t in ys])
    p = torch.cat([t.detach().reshape(-1).float() for t in yp])
    if num_repeats > 1:
        y, p = average_over_repeats(y, num_repeats), average_over_repeats(p, num_repeats)
    return compute_correlations_deferred(y, p) if defer else compute_correlations(y, p)


def predict_repeats(model, pref_module, datas, is_pairwise: bool, use_scales: bool):
    """R loader batches of the SAME images (one per test-time repeat, each with its own patch sample) as ONE forward over
    R*B items.  Returns (q, q_p) pass-major, i.e. exactly the concatenation do_validation builds from R separate passes."""
    datas = list(datas)
    merged = tuple(torch.cat([d[k] for d in datas], dim=0) for k in range(len(datas[0])))
    q, q_p, _ = predict(model, pref_module, merged, is_pairwise, False, use_scales)
    return q, q_p


def do_validation(model, pref_module, device, is_pairwise, loader, num_repeats: int = 1, use_scales: bool = False,
                  output_logger=None, tag: str = "", step: int = 0, defer: bool = False):
    """The scoring part of train.do_validation (train.py:583-644): returns (step, correlations or None).  defer=True: the correlations come
    back as a PendingCorrelations -- the set's device reductions are enqueued, its logistic fit runs on a worker thread while the caller
    starts the next pass; `.result()` gives the dict."""
    y, yp = [], []
    with torch.no_grad():
        model.eval()
        if pref_module is not None:
            pref_module.eval()
        for _ in range(num_repeats):
            for i, batch in enumerate(loader):
                data = get_data_tuple(batch, device)
                q, q_p, _ = predict(model, pref_module, data, is_pairwise, False, use_scales)
                y.append(q)
                yp.append(q_p)
                if output_logger is not None:                        # the score CSV (train.py:627-632) needs the values now
                    output_logger(i, tag, ",".join(str(v) for v in np.array(q_p.detach().cpu())))
                step += 1
    correlations = compute_correlations_cat_flat(y, yp, num_repeats, defer=defer) if y else None
    return step, correlations

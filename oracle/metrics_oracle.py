"""CPU restatement of the reference's validation reductions (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Follows, in numpy / scipy (scipy is the reference's own dependency for these statistics):
  * average_over_repeats            train.py:398-400
  * compute_correlations_cat_flat   train.py:403-409
  * normalize_array                 utils/image_processing/image_tools.py:17-21
  * compute_correlations            utils/misc/correlations.py:21-52 (fit: FitFunction, :57-135, form 1 with 'L1' residuals)

Pinned by tests/golden/validation_metrics.npz, generated from the reference's own functions
(tests/golden/make_golden.py:run_validation_metrics)."""
import numpy as np
import scipy.optimize
import scipy.stats


def average_over_repeats(x, num_repeats):
    return np.asarray(x, dtype=float).reshape(num_repeats, -1).mean(axis=0)


def normalize_array(a):
    b = a - a.min()
    top = b.max()
    return b / top if abs(top) > 1e-6 else b


def logistic5(p, x):
    """correlations.py:124-127: y = p0 (1/2 - 1/(1 + exp(p1 (x - p2) + eps))) + |p3| x + p4, eps = CORRELATIONS_EPS = 1e-6."""
    return p[0] * (0.5 - 1.0 / (1.0 + np.exp(p[1] * (x - p[2]) + 1e-6))) + abs(p[3]) * x + p[4]


def fit_logistic5(source, target):
    guess = (1.0, 1.0, np.median(source), 1.0, np.median(target))
    p = scipy.optimize.leastsq(lambda q, x, y: y - logistic5(q, x), guess, args=(source, target), full_output=True)[0]
    if np.isnan(np.asarray(p)).any():
        raise OverflowError("fit produced NaN")
    return p


def compute_correlations(a, b, normalize=True):
    aa = normalize_array(a) if normalize else a.copy()
    bb = normalize_array(b) if normalize else b.copy()
    out = {
        "SROCC": scipy.stats.spearmanr(aa, bb).correlation,
        "KROCC": scipy.stats.kendalltau(aa, bb).correlation,
        "PLCC_NOFIT": scipy.stats.pearsonr(aa, bb)[0],
        "RMSE_NOFIT": float(np.sqrt(np.mean((aa - bb) ** 2))),
    }
    try:
        bb = logistic5(fit_logistic5(bb, aa), bb)
    except OverflowError:
        pass
    out["PLCC"] = scipy.stats.pearsonr(aa, bb)[0]
    out["RMSE"] = float(np.sqrt(np.mean((aa - bb) ** 2)))
    return out


def compute_correlations_cat_flat(ys, yp, num_repeats=1):
    ys = np.concatenate([np.asarray(t, dtype=float).ravel() for t in ys])
    yp = np.concatenate([np.asarray(t, dtype=float).ravel() for t in yp])
    if num_repeats > 1:
        ys, yp = average_over_repeats(ys, num_repeats), average_over_repeats(yp, num_repeats)
    return compute_correlations(ys, yp)

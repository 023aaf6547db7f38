"""validation-loop reductions.  CPU: the oracle against vectors produced by the reference's own
train.compute_correlations_cat_flat / average_over_repeats.  GPU: the HIP kernels (through the C ABI) against both."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "validation_metrics.npz"))
CASES = ["mos", "ties", "single"]
FIELDS = ["SROCC", "KROCC", "PLCC", "RMSE", "PLCC_NOFIT", "RMSE_NOFIT"]
# tolerances: rank statistics are ratios of exact integers / short fp64 sums; the fitted fields pass through
# scipy.optimize.leastsq (MINPACK), whose iterates move with last-bit differences of the residuals
TOL = {"SROCC": 1e-12, "KROCC": 1e-14, "PLCC_NOFIT": 1e-12, "RMSE_NOFIT": 1e-12, "PLCC": 1e-7, "RMSE": 1e-6}


def batches(name):
    q, pred, reps, bs = G[name + "_q"], G[name + "_pred"], int(G[name + "_reps"]), int(G[name + "_bs"])
    n = q.size
    ys = [q[i:i + bs] for _ in range(reps) for i in range(0, n, bs)]
    yp = [pred[r * n + i:r * n + min(i + bs, n)] for r in range(reps) for i in range(0, n, bs)]
    return ys, yp, reps


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_vectors(name):
    ys, yp, reps = batches(name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = MO.compute_correlations_cat_flat(ys, yp, reps)
    for f in FIELDS:
        assert abs(got[f] - float(G[f"{name}_{f}"])) <= TOL[f], (f, got[f], float(G[f"{name}_{f}"]))
    if reps > 1:
        assert np.array_equal(MO.average_over_repeats(G[name + "_pred"], reps), G[name + "_mean"])


def test_product_path_refuses_cpu_tensors():
    from vtamiq_amd import validate
    with pytest.raises(RuntimeError):
        validate.average_over_repeats(torch.zeros(8), 2)
    with pytest.raises(RuntimeError):
        validate.compute_correlations(torch.zeros(8), torch.ones(8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_metrics_match_reference_vectors(name):
    from vtamiq_amd import validate
    ys, yp, reps = batches(name)
    dev = [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in ys], [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in yp]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = validate.compute_correlations_cat_flat(dev[0], dev[1], reps)
    for f in FIELDS:
        assert abs(got[f] - float(G[f"{name}_{f}"])) <= TOL[f], (f, got[f], float(G[f"{name}_{f}"]))
    # the deferred form (fit on a worker thread, nothing waits for the device until result()): the same numbers, bit for bit
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pend = validate.compute_correlations_cat_flat(dev[0], dev[1], reps, defer=True)
        later = pend.result()
    assert all((later[f] == got[f]) or (np.isnan(later[f]) and np.isnan(got[f])) for f in FIELDS) and pend.done()
    if reps > 1:                                             # bit-exact: same summation order as numpy's axis-0 reduction
        m = validate.average_over_repeats(torch.from_numpy(G[name + "_pred"]).cuda(), reps).cpu().numpy()
        assert np.array_equal(m, G[name + "_mean"])


@pytest.mark.gpu
def test_hip_metrics_large_random_vs_oracle():
    """N = 5000 with heavy ties: exact Kendall pair counts and ranks against scipy (the oracle)."""
    from vtamiq_amd import validate
    import scipy.stats
    rng = np.random.default_rng(5)
    a = np.round(rng.uniform(0, 5, 5000), 1)
    b = np.round(0.6 * a + rng.standard_normal(5000), 1)
    got = validate.compute_correlations(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), normalize=False)
    assert abs(got["KROCC"] - scipy.stats.kendalltau(a, b).correlation) <= 1e-14
    assert abs(got["SROCC"] - scipy.stats.spearmanr(a, b).correlation) <= 1e-12
    assert abs(got["PLCC_NOFIT"] - scipy.stats.pearsonr(a, b)[0]) <= 1e-12
    const = validate.compute_correlations(torch.ones(16).cuda(), torch.arange(16.).cuda())
    assert np.isnan(const["KROCC"])


@pytest.mark.gpu
def test_predict_repeats_equals_separate_passes():
    """R repeats as ONE forward give the pass-major concatenation of R separate forwards, bit for bit."""
    from vtamiq_amd import VTAMIQ, synth, validate, predict
    m = VTAMIQ(vit_config=dict(variant="ViT-B16", num_keep_layers=2))
    sd = synth.make_state_dict(m.spec, 3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda().eval()
    B, N, R = 6, 40, 3                                   # the logistic fit needs >= 5 images (MINPACK: M >= 5 parameters)
    datas = []
    for r in range(R):
        patches, pos, _ = synth.make_inputs(m.spec, B, N, 100 + r)
        datas.append((torch.linspace(0.1, 0.9, B).cuda(), torch.from_numpy(patches).cuda(), torch.from_numpy(pos).cuda(),
                      torch.full((B,), -1, dtype=torch.int32).cuda()))
    with torch.no_grad():
        q, qp = validate.predict_repeats(m, None, datas, False, False)
        sep = torch.cat([predict(m, None, d, False, False, False)[1] for d in datas])
    assert torch.equal(qp, sep)
    mean = validate.average_over_repeats(qp, R)
    assert torch.allclose(mean, qp.double().reshape(R, B).mean(0), rtol=0, atol=1e-15)
    step, corr = validate.do_validation(m, None, torch.device("cuda"), False, [tuple(t.cpu() for t in d) for d in datas[:1]], num_repeats=2)
    assert step == 2 and set(corr) == {"SROCC", "KROCC", "PLCC", "RMSE", "PLCC_NOFIT", "RMSE_NOFIT"}
    step2, pend = validate.do_validation(m, None, torch.device("cuda"), False, [tuple(t.cpu() for t in d) for d in datas[:1]], num_repeats=2, defer=True)
    later = pend.result()
    assert step2 == 2 and all((later[k] == corr[k]) or (np.isnan(later[k]) and np.isnan(corr[k])) for k in corr)
